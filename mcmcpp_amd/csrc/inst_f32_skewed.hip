// kernels for ParamType = float, Calculator = SkewedGaussian2DFn
#define INST_T float
#define INST_CALC SkewedGaussian2DFn
#define INST_NAME launch_table_f32_skewed
#define INST_ONLY_LPW1 1
#include "instances.inc"
