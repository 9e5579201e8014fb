// kernels for ParamType = double, Calculator = SkewedGaussian2DFn
#define INST_T double
#define INST_CALC SkewedGaussian2DFn
#define INST_NAME launch_table_f64_skewed
#define INST_ONLY_LPW1 1
#include "instances.inc"
