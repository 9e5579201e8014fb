// kernels for ParamType = float, Calculator = IsoGaussianFn
#define INST_T float
#define INST_CALC IsoGaussianFn
#define INST_NAME launch_table_f32_iso
#define INST_ONLY_LPW1 0
#include "instances.inc"
