// kernels for ParamType = float, Calculator = DenseGaussianFn
#define INST_T float
#define INST_CALC DenseGaussianFn
#define INST_NAME launch_table_f32_dense
#define INST_ONLY_LPW1 0
#include "instances.inc"
