// kernels for ParamType = double, Calculator = RosenbrockFn
#define INST_T double
#define INST_CALC RosenbrockFn
#define INST_NAME launch_table_f64_rosenbrock
#define INST_ONLY_LPW1 0
#include "instances.inc"
