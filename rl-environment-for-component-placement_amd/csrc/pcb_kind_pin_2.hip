// pcb_kind_pin_2.hip -- kernels of the pin environment, part 2 (pcb_kind.inc lists the parts; one translation unit each: they compile in parallel)
#include <hip/hip_runtime.h>
#include "pcbenv.h"
#define PCB_KIND PCBENV_PIN
#define PCB_KIND_NAME pin
#define PCB_PART 2
#include "pcb_kind.inc"
