// pcb_kind_square.hip -- kernels of the square environment (one translation unit per kind: they compile in parallel)
#include <hip/hip_runtime.h>
#include "pcbenv.h"
#define PCB_KIND PCBENV_SQUARE
#define PCB_KIND_NAME square
#include "pcb_kind.inc"
