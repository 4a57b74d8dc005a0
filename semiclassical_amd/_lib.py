"""ctypes binding of libsemiclassical_hip.so (declared in include/semiclassical_hip.h).

There is no fallback: if the shared library is missing or a symbol cannot be
resolved the import fails loudly -- the engine has no CPU path.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SC_LIB_PATH") or os.path.join(HERE, "libsemiclassical_hip.so")

c_double_p = C.c_void_p      # device pointers travel as plain integers


class sc_potential(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32),
                ("par0", c_double_p), ("par1", c_double_p), ("par2", c_double_p),
                ("scalar0", C.c_double), ("inv_mass", c_double_p),
                ("lin_prop", c_double_p), ("lin_dt", C.c_double)]


class sc_state(C.Structure):
    _fields_ = [("n", C.c_int64), ("dim", C.c_int32), ("mono_layout", C.c_int32),
                ("qp", c_double_p), ("act", c_double_p), ("mono", c_double_p),
                ("c2", c_double_p), ("sgn", c_double_p), ("work", c_double_p), ("flags", C.c_void_p)]


class sc_hk_consts(C.Structure):
    _fields_ = [("dim", C.c_int32), ("dprime", C.c_int32), ("diag", C.c_int32), ("real_lr", C.c_int32),
                ("st", c_double_p), ("si", c_double_p),
                ("L1", c_double_p), ("L2", c_double_p), ("R1", c_double_p), ("R2", c_double_p)]


class sc_overlap_consts(C.Structure):
    _fields_ = [("dim", C.c_int32), ("diag", C.c_int32),
                ("A", c_double_p), ("B", c_double_p), ("C", c_double_p),
                ("qk", c_double_p), ("pk", c_double_p), ("fac", C.c_double)]


class sc_nac_consts(C.Structure):
    _fields_ = [("dim", C.c_int32), ("_pad", C.c_int32),
                ("rn", c_double_p), ("gn", c_double_p), ("q0", c_double_p), ("p0", c_double_p),
                ("p0n1", C.c_double), ("n2", C.c_double)]


class sc_wm_consts(C.Structure):
    _fields_ = [("dim", C.c_int32), ("dprime", C.c_int32),
                ("U", c_double_p), ("Gt", c_double_p), ("G0", c_double_p), ("iGi0", c_double_p),
                ("S", c_double_p), ("Cqq", c_double_p), ("Cst", c_double_p), ("Bq", c_double_p),
                ("q0", c_double_p), ("p0", c_double_p),
                ("n1", c_double_p), ("s_n1", c_double_p), ("w_n1", c_double_p),
                ("inv_scale_a", C.c_double), ("inv_two_pi", C.c_double), ("pre", C.c_double),
                ("p0n1", C.c_double), ("n2", C.c_double),
                ("detA", c_double_p), ("detM", c_double_p), ("sgnA", c_double_p), ("sgnM", c_double_p),
                ("pre_coef", C.c_double), ("coef_out", c_double_p), ("cqq_out", c_double_p), ("dvec_out", c_double_p),
                ("scratch", c_double_p), ("scratch_bytes", C.c_int64), ("flags", C.c_void_p),
                ("nac_traj", c_double_p)]


class sc_gdml_model(C.Structure):
    _fields_ = [("n_atoms", C.c_int32), ("n_desc", C.c_int32), ("n_train", C.c_int32), ("_pad", C.c_int32),
                ("xs_train", c_double_p), ("jx_alphas", c_double_p), ("pair_k", C.c_void_p), ("pair_l", C.c_void_p),
                ("q", C.c_double), ("c", C.c_double), ("std", C.c_double), ("origin", C.c_double),
                ("inv_mass", c_double_p)]


class sc_multi_scratch(C.Structure):
    _fields_ = [("work", c_double_p), ("qp_mid", c_double_p), ("act_mid", c_double_p), ("c2_mid", c_double_p),
                ("sgn_mid", c_double_p), ("unrepaired", C.c_void_p)]


class sc_dense_scratch(C.Structure):
    _fields_ = [("hess", c_double_p), ("kprev", c_double_p), ("ksum", c_double_p), ("ssum", c_double_p)]


SC_POT_MORSE, SC_POT_HARMONIC_SEP, SC_POT_EPS_MORSE, SC_POT_HARMONIC_DENSE = 1, 2, 3, 4
SC_MONO_ROWMAJOR, SC_MONO_TILED16 = 0, 1

# every symbol include/semiclassical_hip.h declares: name -> (restype, argtypes)
P = C.POINTER
ABI_VERSION = 17              # = SC_ABI_VERSION of include/semiclassical_hip.h these declarations were written against
STRUCTS = (sc_potential, sc_state, sc_hk_consts, sc_overlap_consts, sc_nac_consts, sc_wm_consts, sc_gdml_model,
           sc_dense_scratch, sc_multi_scratch)

SIGNATURES = {
    "sc_version": (C.c_int, []),
    "sc_abi_version": (C.c_int, []),
    "sc_struct_size": (C.c_int, [C.c_char_p]),
    "sc_tuning_build": (C.c_int, []),
    "sc_last_error": (C.c_char_p, []),
    "sc_step_grid": (C.c_int, [C.c_int64, C.c_int32]),
    "sc_correlate_grid": (C.c_int, [C.c_int64, C.c_int32]),
    "sc_mono_convert": (C.c_int, [P(sc_state), C.c_int32, C.c_void_p]),
    "sc_state_from_reference": (C.c_int, [c_double_p, P(sc_state), C.c_void_p]),
    "sc_state_to_reference": (C.c_int, [P(sc_state), c_double_p, C.c_void_p]),
    "sc_sample_initial": (C.c_int, [P(sc_state), c_double_p, c_double_p, C.c_int32, C.c_double, C.c_uint64, C.c_uint64,
                                    C.c_int64, C.c_int32, c_double_p, c_double_p, c_double_p, C.c_void_p]),
    "sc_hk_step": (C.c_int, [P(sc_potential), P(sc_state), P(sc_hk_consts), C.c_double, C.c_int32,
                             c_double_p, C.c_void_p]),
    "sc_hk_step_multi_supported": (C.c_int, [P(sc_potential), P(sc_state), P(sc_hk_consts)]),
    "sc_hk_step_multi": (C.c_int, [P(sc_potential), P(sc_state), P(sc_hk_consts), P(sc_multi_scratch), C.c_double, c_double_p,
                                   C.c_void_p]),
    "sc_overlap": (C.c_int, [P(sc_overlap_consts), c_double_p, C.c_int64, c_double_p, C.c_void_p]),
    "sc_nac_initial": (C.c_int, [P(sc_nac_consts), c_double_p, C.c_int64, c_double_p, C.c_void_p]),
    "sc_hk_correlate": (C.c_int, [P(sc_state), P(sc_overlap_consts), P(sc_nac_consts), c_double_p, c_double_p,
                                  c_double_p, C.c_double, c_double_p, c_double_p, c_double_p, C.c_void_p]),
    "sc_hk_run_slots": (C.c_int, [C.c_int64, C.c_int32]),
    "sc_hk_run_supported": (C.c_int, [P(sc_potential), P(sc_hk_consts), P(sc_overlap_consts)]),
    "sc_hk_run": (C.c_int, [P(sc_potential), P(sc_state), P(sc_hk_consts), P(sc_overlap_consts), P(sc_nac_consts), c_double_p,
                            c_double_p, c_double_p, C.c_double, C.c_double, C.c_int32, c_double_p, c_double_p, c_double_p,
                            C.c_void_p]),
    "sc_mono_similarity": (C.c_int, [P(sc_state), c_double_p, c_double_p, C.c_void_p]),
    "sc_hk_run_modal_supported": (C.c_int, [P(sc_potential), P(sc_hk_consts), P(sc_overlap_consts)]),
    "sc_hk_run_modal": (C.c_int, [P(sc_potential), P(sc_state), P(sc_hk_consts), P(sc_overlap_consts), P(sc_nac_consts), c_double_p,
                                  c_double_p, c_double_p, C.c_double, C.c_double, C.c_int32, c_double_p, c_double_p, c_double_p,
                                  c_double_p, C.c_void_p]),
    "sc_energy_guard": (C.c_int, [c_double_p, C.c_int32, C.c_double, c_double_p, C.c_void_p]),
    "sc_wm_grid": (C.c_int, [C.c_int64, C.c_int32]),
    "sc_wm_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "sc_wm_correlate": (C.c_int, [P(sc_state), P(sc_wm_consts), c_double_p, c_double_p, C.c_double, C.c_int32,
                                  C.c_int32, c_double_p, c_double_p, c_double_p, C.c_void_p]),
    "sc_wm_grid_sum": (C.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64, C.c_int32, c_double_p,
                                 C.c_int32, c_double_p, C.c_void_p]),
    "sc_wm_pair_sum_tiles": (C.c_int64, [C.c_int64]),
    "sc_wm_pair_sum": (C.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                 C.c_int64, C.c_int32, C.c_int32, c_double_p, C.c_void_p]),
    "sc_gdml_row_len": (C.c_int, [C.c_int32]),
    "sc_gdml_eval": (C.c_int, [P(sc_gdml_model), c_double_p, C.c_int64, c_double_p, c_double_p, c_double_p,
                               C.c_void_p]),
    "sc_dense_grid": (C.c_int, [C.c_int64]),
    "sc_gdml_stage": (C.c_int, [P(sc_gdml_model), P(sc_state), P(sc_dense_scratch), C.c_double, C.c_int32,
                                c_double_p, C.c_void_p]),
    "sc_stage_point": (C.c_int, [P(sc_state), P(sc_dense_scratch), C.c_double, C.c_int32, c_double_p, C.c_void_p]),
    "sc_stage_consume": (C.c_int, [P(sc_state), P(sc_dense_scratch), c_double_p, c_double_p, c_double_p, C.c_double,
                                   C.c_int32, c_double_p, C.c_void_p]),
    "sc_dense_mono_scratch_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "sc_dense_mono_step": (C.c_int, [P(sc_state), P(sc_hk_consts), c_double_p, c_double_p, c_double_p, C.c_double,
                                     C.c_int32, C.c_void_p]),
    "sc_pair_sum_tiles": (C.c_int64, [C.c_int64]),
    "sc_pair_sum_rect_tiles": (C.c_int64, [C.c_int64, C.c_int64]),
    "sc_pair_sum_rect": (C.c_int, [c_double_p, c_double_p, C.c_int32, c_double_p, c_double_p, C.c_int32, c_double_p, c_double_p,
                                   c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64, C.c_int64, c_double_p, C.c_void_p]),
    "sc_wm_pair_sum_rect_tiles": (C.c_int64, [C.c_int64, C.c_int64]),
    "sc_wm_pair_sum_rect": (C.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64, c_double_p, c_double_p, c_double_p,
                                      c_double_p, c_double_p, c_double_p, C.c_int64, c_double_p, C.c_int32, C.c_int32, c_double_p,
                                      C.c_void_p]),
    "sc_pair_sum": (C.c_int, [c_double_p, c_double_p, C.c_int32, c_double_p, c_double_p, C.c_int32, c_double_p,
                              c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64, c_double_p, C.c_void_p]),
    "sc_hk_step_diag": (C.c_int, [P(sc_potential), P(sc_state), P(sc_hk_consts), c_double_p, C.c_double, C.c_int32,
                                  c_double_p, C.c_void_p]),
    "sc_grid_sum": (C.c_int, [c_double_p, c_double_p, c_double_p, c_double_p, C.c_int64, C.c_int32, c_double_p,
                              c_double_p, C.c_int32, C.c_double, c_double_p, C.c_void_p]),
    "sc_reduce_slot_at": (C.c_int, [c_double_p, C.c_int32, c_double_p, C.c_void_p, C.c_void_p]),
    "sc_reduce_slot": (C.c_int, [c_double_p, C.c_int32, c_double_p, C.c_int32, C.c_double, c_double_p,
                                 C.c_void_p]),
    "sc_comm_available": (C.c_int, []),
    "sc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "sc_comm_init": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, P(C.c_void_p)]),
    "sc_comm_rank_count": (C.c_int, [C.c_void_p, P(C.c_int32), P(C.c_int32)]),
    "sc_comm_destroy": (C.c_int, [C.c_void_p]),
    "sc_flush_allreduce": (C.c_int, [c_double_p, C.c_int64, C.c_void_p, C.c_void_p]),
}


class EngineError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m semiclassical_amd.build` "
            "(hipcc --offload-arch=gfx950).  semiclassical_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    # ABI guard: a stale or foreign build must fail here, not run with shifted struct fields
    if lib.sc_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {lib.sc_abi_version()}, these bindings need {ABI_VERSION}: "
                          "rebuild with `python -m semiclassical_amd.build --force`")
    for st in STRUCTS:
        have = lib.sc_struct_size(st.__name__.encode())
        if have != C.sizeof(st):
            raise ImportError(f"{LIB_PATH}: sizeof({st.__name__}) is {have} in the library, {C.sizeof(st)} in the bindings")
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise EngineError(f"semiclassical_hip error {rc}: {lib.sc_last_error().decode()}")


def ptr(t):
    """device pointer of a torch tensor (or NULL)"""
    return None if t is None else C.c_void_p(t.data_ptr())
