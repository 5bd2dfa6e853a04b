// sa_compact.hip -- placeholder translation unit, filled in below.
#include <hip/hip_runtime.h>
