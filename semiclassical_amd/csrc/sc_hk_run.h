// Arguments shared by the whole-loop kernels behind sc_hk_run (sc_hk_run_sep16.hip: separable potentials;
// sc_hk_run_lin.hip: constant dense Hessian).
#pragma once
#include "sc_common.h"

struct RunArgs {
    StepArgs step;              // potential, state, prefactor constants, dt
    sc_overlap_consts oc;       // <q_t, p_t, Gamma_t | q0, p0, Gamma_0>
    sc_nac_consts nc;
    int has_nac;
    const double *vi, *probi, *nacq;
    double mc_norm;
    int nsteps;
    double *partials;           // [nsteps][slots][5]: Re C, Im C, Re k, Im k, sum of (T+V) at the k4 stage; zeroed by the caller
    int slots;                  // 4 * gridDim.x
    const double *mode_prop;    // sc_hk_run_modal: [D][4] per-mode step matrices (the blocks are in normal-mode coordinates), else NULL
};

// sc_hk_run_lin.hip: 1 = the shape (D, d', diagonal widths) is instantiated (and, with launch != 0, was launched), 0 = not,
// < 0 on error.  The caller guarantees SC_POT_HARMONIC_DENSE, pot.lin_prop built for a.step.dt, row-major blocks.
int sc_launch_run_lin(const RunArgs &a, int grid, hipStream_t s, int launch);
