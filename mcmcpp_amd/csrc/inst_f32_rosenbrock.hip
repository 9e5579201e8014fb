// kernels for ParamType = float, Calculator = RosenbrockFn
#define INST_T float
#define INST_CALC RosenbrockFn
#define INST_NAME launch_table_f32_rosenbrock
#define INST_ONLY_LPW1 0
#include "instances.inc"
