// pcb_kind_spatial_1.hip -- kernels of the spatial environment, part 1 (pcb_kind.inc lists the parts; one translation unit each: they compile in parallel)
#include <hip/hip_runtime.h>
#include "pcbenv.h"
#define PCB_KIND PCBENV_SPATIAL
#define PCB_KIND_NAME spatial
#define PCB_PART 1
#include "pcb_kind.inc"
