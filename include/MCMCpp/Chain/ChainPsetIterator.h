/*
 * Chain/ChainPsetIterator.h -- walks the chain one parameter set (one walker of one step) at a time.
 *
 * Contract of the reference's iterator (/root/reference/MCMCpp/Chain/ChainPsetIterator.h:55-117): `*it`
 * is a ParamType* to D contiguous values; ++ visits the next walker of the step and then the first
 * walker of the next step, regardless of step boundaries.  The reference's ++ visits one phantom cell at
 * a full-block boundary and its -- tests the wrong link (ChainPsetIterator.h:131-142,154); indexing by a
 * global 64-bit parameter-set number has neither problem.
 */
#ifndef MCMCPP_CHAIN_CHAINPSETITERATOR_H
#define MCMCPP_CHAIN_CHAINPSETITERATOR_H

#include <cstdint>

namespace MCMC
{
namespace Chain
{
template <class ParamType>
class Chain;

template <class ParamType>
class ChainPsetIterator
{
public:
    ChainPsetIterator(Chain<ParamType>* owner, std::int64_t psetIndex) : chain(owner), index(psetIndex) {}

    bool operator==(const ChainPsetIterator& rhs) const { return chain == rhs.chain && index == rhs.index; }
    bool operator!=(const ChainPsetIterator& rhs) const { return !(*this == rhs); }

    ChainPsetIterator& operator++()
    {
        const std::int64_t last = chain->getStoredStepCount() * chain->getWalkerCount();
        if (index < last) ++index;
        return *this;
    }
    ChainPsetIterator& operator--()
    {
        if (index > 0) --index;
        return *this;
    }

    /// The D parameters of this walker at this step.
    ParamType* operator*() const
    {
        const std::int64_t walkers = chain->getWalkerCount();
        return chain->stepPtr(index / walkers) + (index % walkers) * chain->getCellsPerWalker();
    }

private:
    Chain<ParamType>* chain;
    std::int64_t index;
};

}  // namespace Chain
}  // namespace MCMC
#endif  // MCMCPP_CHAIN_CHAINPSETITERATOR_H
