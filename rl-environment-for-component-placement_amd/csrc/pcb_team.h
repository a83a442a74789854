// pcb_team.h -- everything a TEAM of TN threads does for one environment, as static members of Team<TN>.
// Part of libpcbenv.so (CDNA4 / gfx950 only).
//
// A team is the set of threads that work on one environment: one wavefront (TN = 64) for grids up to 64 x 64, four
// (TN = 256) for the 128 x 128 spatial configuration -- and, since round 3, four for the environments of a launch that
// are certain to end their episode in it, next to one-wavefront teams for the others in the same kernel (k_step_mixed).
// The team size decides the lane stride of every loop, whether a phase boundary needs an s_barrier (a one-wavefront
// team's LDS traffic executes in program order) and how work is dealt to wavefronts, so it is a compile-time
// property: the sections below are textually included inside the class template, where NT is TN.  They hold the
// device functions only; the __global__ kernels that pick teams are in pcb_kernels.h.
#pragma once
#include "pcb_device.h"

#define NT TN  // inside Team<TN> only (undefined again below)
template <int TN> struct Team {
    static_assert(TN == 64 || TN == 256, "one or four wavefronts per environment");
#include "pcb_team_io.h"   // LDS barrier, any(), plane emission, window fold
#include "pcb_reward.h"    // centroid routes, intersection count, wirelength
#include "pcb_beam.h"      // beam-search routes
#include "pcb_observe.h"   // state staging, mask + observation emission, pin_grid, features, terminal reward
#include "pcb_sampler.h"   // uniform legal-action draw
#include "pcb_reset.h"     // reset from the instance queue
#include "pcb_step.h"      // transition, run_env
};
#undef NT
