/*
 * Utility/UserOjbectsTest.h -- compile-time checks of the user-supplied plug-in classes (file name kept
 * from the reference so that existing include lines resolve: /root/reference/MCMCpp/Utility/UserOjbectsTest.h).
 *
 * The plug-in boundary of the reference is a set of member-function signatures
 * (UserOjbectsTest.h:132-151):
 *   Calculator      ParamType calcLogPostProb(ParamType*)
 *   distribution    ParamType operator()(ParamType)
 *   PostStepAction  void performAction(const StepIterator&, const StepIterator&)
 *   Mover           void updateWalker(Walker&, Walker*, int, bool)
 * The same four traits are provided (same names), built on one generic detector.  A fifth trait,
 * CheckDeviceCalculator, is new: it asks for the device identity a Calculator needs on the MI355X path.
 */
#ifndef MCMCPP_UTILITY_USEROBJECTTEST_H
#define MCMCPP_UTILITY_USEROBJECTTEST_H

#include <type_traits>
#include <utility>

namespace MCMC
{
namespace Utility
{
namespace Detail
{
template <class...>
struct MakeVoid
{
    typedef void type;
};

// Primary: the expression does not compile for these types.
template <class Enable, template <class...> class Expr, class... Args>
struct Detect : std::false_type
{
    typedef void type;
};
template <template <class...> class Expr, class... Args>
struct Detect<typename MakeVoid<Expr<Args...> >::type, Expr, Args...> : std::true_type
{
    typedef Expr<Args...> type;
};

template <class C, class A0>
using CalcLogPostProbExpr = decltype(std::declval<C&>().calcLogPostProb(std::declval<A0>()));
template <class C, class A0>
using FunctorExpr = decltype(std::declval<C&>()(std::declval<A0>()));
template <class C, class A0, class A1>
using PerformActionExpr = decltype(std::declval<C&>().performAction(std::declval<A0>(), std::declval<A1>()));
template <class C, class A0, class A1, class A2, class A3>
using UpdateWalkerExpr =
    decltype(std::declval<C&>().updateWalker(std::declval<A0>(), std::declval<A1>(), std::declval<A2>(), std::declval<A3>()));
template <class C>
using DeviceIdentityExpr = decltype(static_cast<int>(C::hipCalcId) + std::declval<const C&>().hipParamCount() +
                                    (std::declval<const C&>().hipParams() ? 1 : 0));

template <bool Found, class Got, class Want>
struct ReturnsExactly : std::false_type
{
};
template <class Got, class Want>
struct ReturnsExactly<true, Got, Want> : std::is_same<Got, Want>
{
};
}  // namespace Detail

template <class TestClass, class RetVal, class Arg0>
struct CheckCalcLogPostProb
    : Detail::ReturnsExactly<Detail::Detect<void, Detail::CalcLogPostProbExpr, TestClass, Arg0>::value,
                             typename Detail::Detect<void, Detail::CalcLogPostProbExpr, TestClass, Arg0>::type, RetVal>
{
};

template <class TestClass, class RetVal, class Arg0>
struct CheckFunctor
    : Detail::ReturnsExactly<Detail::Detect<void, Detail::FunctorExpr, TestClass, Arg0>::value,
                             typename Detail::Detect<void, Detail::FunctorExpr, TestClass, Arg0>::type, RetVal>
{
};

template <class TestClass, class RetVal, class Arg0, class Arg1>
struct CheckPerformAction
    : Detail::ReturnsExactly<Detail::Detect<void, Detail::PerformActionExpr, TestClass, Arg0, Arg1>::value,
                             typename Detail::Detect<void, Detail::PerformActionExpr, TestClass, Arg0, Arg1>::type, RetVal>
{
};

template <class TestClass, class RetVal, class Arg0, class Arg1, class Arg2, class Arg3>
struct CheckCalcUpdateWalker
    : Detail::ReturnsExactly<Detail::Detect<void, Detail::UpdateWalkerExpr, TestClass, Arg0, Arg1, Arg2, Arg3>::value,
                             typename Detail::Detect<void, Detail::UpdateWalkerExpr, TestClass, Arg0, Arg1, Arg2, Arg3>::type,
                             RetVal>
{
};

/// True when the Calculator also names a device functor: `static const int hipCalcId`,
/// `const ParamType* hipParams() const`, `int hipParamCount() const` (see Device/Calculators.h).
template <class TestClass>
struct CheckDeviceCalculator : Detail::Detect<void, Detail::DeviceIdentityExpr, TestClass>
{
};

}  // namespace Utility
}  // namespace MCMC
#endif  // MCMCPP_UTILITY_USEROBJECTTEST_H
