// pcb_launch.h -- what the host side (pcbenv_kernels.hip) and the per-kind kernel translation units (pcb_kind_*.hip)
// share: one launch entry per environment kind, so that the kernel instantiations of the four kinds compile in parallel.
#pragma once
#include "pcb_device.h"

struct StepLaunch {
    DevParams d;
    int *actions;
    int fmt, sampled;
    u64 seed, first_env, step_index;
    int num_steps;
    int threads;        // threads per environment of the plain kernels (64 / 256)
    bool routes, traj;  // beam / both routes compiled in; trajectory layout or persistent rollout
    hipStream_t stream;
};
struct ResetLaunch { DevParams d; const unsigned char *mask; int threads; hipStream_t stream; };

#define PCB_DECLARE_KIND(name) int pcb_launch_step_##name(const StepLaunch &a); int pcb_launch_reset_##name(const ResetLaunch &a);
PCB_DECLARE_KIND(square) PCB_DECLARE_KIND(rect) PCB_DECLARE_KIND(pin) PCB_DECLARE_KIND(spatial)
