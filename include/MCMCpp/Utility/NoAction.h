/*
 * Utility/NoAction.h -- the do-nothing PostStepAction (default third template argument of the samplers,
 * as in /root/reference/MCMCpp/Utility/NoAction.h:36-45).
 */
#ifndef MCMCPP_UTILITY_NOACTION_H
#define MCMCPP_UTILITY_NOACTION_H

#include "../Chain/ChainStepIterator.h"

namespace MCMC
{
namespace Utility
{
template <class ParamType>
class NoAction
{
public:
    void performAction(const Chain::ChainStepIterator<ParamType>&, const Chain::ChainStepIterator<ParamType>&) {}
};
}  // namespace Utility
}  // namespace MCMC
#endif  // MCMCPP_UTILITY_NOACTION_H
