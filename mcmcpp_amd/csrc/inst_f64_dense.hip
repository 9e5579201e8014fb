// kernels for ParamType = double, Calculator = DenseGaussianFn
#define INST_T double
#define INST_CALC DenseGaussianFn
#define INST_NAME launch_table_f64_dense
#define INST_ONLY_LPW1 0
#include "instances.inc"
