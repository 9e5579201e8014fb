// kernels for ParamType = double, Calculator = IsoGaussianFn
#define INST_T double
#define INST_CALC IsoGaussianFn
#define INST_NAME launch_table_f64_iso
#define INST_ONLY_LPW1 0
#include "instances.inc"
