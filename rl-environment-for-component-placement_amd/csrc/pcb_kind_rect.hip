// pcb_kind_rect.hip -- kernels of the rect environment (one translation unit per kind: they compile in parallel)
#include <hip/hip_runtime.h>
#include "pcbenv.h"
#define PCB_KIND PCBENV_RECT
#define PCB_KIND_NAME rect
#include "pcb_kind.inc"
