// pcb_kind_pin_1.hip -- kernels of the pin environment, part 1 (pcb_kind.inc lists the parts; one translation unit each: they compile in parallel)
#include <hip/hip_runtime.h>
#include "pcbenv.h"
#define PCB_KIND PCBENV_PIN
#define PCB_KIND_NAME pin
#define PCB_PART 1
#include "pcb_kind.inc"
