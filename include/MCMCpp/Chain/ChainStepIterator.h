/*
 * Chain/ChainStepIterator.h -- walks the chain one stored ensemble step at a time.
 *
 * Same contract as the reference's iterator (/root/reference/MCMCpp/Chain/ChainStepIterator.h:61-128):
 * `*it` is a ParamType* to the W*D contiguous values of a step (walker-major), ++/-- move by one step,
 * +=/-= by many, and movement saturates at the ends of the chain (begin stays begin, end stays end)
 * instead of running off.  Addressing is a 64-bit global step index into the owning Chain.
 */
#ifndef MCMCPP_CHAIN_CHAINSTEPITERATOR_H
#define MCMCPP_CHAIN_CHAINSTEPITERATOR_H

#include <cstdint>

namespace MCMC
{
namespace Chain
{
template <class ParamType>
class Chain;

template <class ParamType>
class ChainStepIterator
{
public:
    ChainStepIterator(Chain<ParamType>* owner, std::int64_t stepIndex) : chain(owner), index(stepIndex) {}

    bool operator==(const ChainStepIterator& rhs) const { return chain == rhs.chain && index == rhs.index; }
    bool operator!=(const ChainStepIterator& rhs) const { return !(*this == rhs); }

    ChainStepIterator& operator++() { return (*this) += 1; }
    ChainStepIterator& operator--() { return (*this) -= 1; }
    ChainStepIterator& operator+=(std::int64_t steps)
    {
        const std::int64_t last = chain->getStoredStepCount();
        index = (steps >= last - index) ? last : index + steps;
        return *this;
    }
    ChainStepIterator& operator-=(std::int64_t steps)
    {
        index = (steps >= index) ? 0 : index - steps;
        return *this;
    }

    /// All walkers of this step: W*D values, walker w at [w*D, (w+1)*D).
    ParamType* operator*() const { return chain->stepPtr(index); }
    std::int64_t stepIndex() const { return index; }

private:
    Chain<ParamType>* chain;
    std::int64_t index;
};

}  // namespace Chain
}  // namespace MCMC
#endif  // MCMCPP_CHAIN_CHAINSTEPITERATOR_H
