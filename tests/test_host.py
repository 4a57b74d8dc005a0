"""CPU-only checks of the host side: setup algebra against the reference's golden intermediates, the
C-ABI library (loads, exports every symbol of include/semiclassical_hip.h -- no compute calls), readers."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests import cases

torch.set_default_dtype(torch.float64)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["hk_as5_chi002", "hk_methylium", "hk_1d"])
def test_sampling_matrices_match_reference(name):
    from semiclassical_amd import hostmath
    g = cases.load(name)
    U, iGi0, iLz, detLz, dprime = hostmath.sampling_matrices(cases.T(g["Gamma_i"]), cases.T(g["Gamma_0"]))
    assert cases.rel_err(U.numpy(), g["U"]) < 1e-14
    assert cases.rel_err(iGi0.numpy(), g["iGi0"]) < 1e-13
    assert dprime == g["U"].shape[1]


def test_prefactor_constants_reproduce_reference_matrix():
    """1/2[L1 Mqq R1 + L2 Mpp R2 - i L1 Mqp R2 + i L2 Mpq R1] == U^T mat U of the reference (via the oracle)"""
    from oracle import sc_oracle as orc
    from semiclassical_amd import hostmath
    g = cases.load("hk_methylium")
    Gi, Gt = cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"])
    pre = hostmath.PrefactorConstants(Gi, Gt, cases.T(g["U"]))
    assert not pre.diag and pre.dprime == 6
    ref = cases.oracle_propagator(g)
    ref.y = cases.T(g["y_10"])
    ref._prefactor()
    Mqq, Mqp, Mpq, Mpp = (X[:, :, 3].type(torch.complex128) for X in ref.monodromy_matrices())
    mat = 0.5 * (pre.L1 @ Mqq @ pre.R1 + pre.L2 @ Mpp @ pre.R2
                 - 1j * pre.L1 @ Mqp @ pre.R2 + 1j * pre.L2 @ Mpq @ pre.R1)
    assert abs(torch.linalg.det(mat) - ref.c2[3]) / abs(ref.c2[3]) < 1e-10


def test_diag_detection():
    from semiclassical_amd import hostmath
    g = cases.load("hk_as5_chi002")
    pre = hostmath.PrefactorConstants(cases.T(g["Gamma_i"]), cases.T(g["Gamma_t"]), cases.T(g["U"]))
    assert pre.diag and torch.allclose(pre.st ** 2, torch.diagonal(cases.T(g["Gamma_t"])))
    oc = hostmath.OverlapConstants(cases.T(g["Gamma_t"]), cases.T(g["Gamma_0"]))
    assert oc.diag and oc.rank == 5


def test_overlap_constants_match_oracle():
    from oracle import sc_oracle as orc
    from semiclassical_amd import hostmath
    g = cases.load("hk_methylium")
    Gt, G0 = cases.T(g["Gamma_t"]), cases.T(g["Gamma_0"])
    oc, ref = hostmath.OverlapConstants(Gt, G0), orc.OverlapOracle(Gt, G0)
    assert not oc.diag and oc.rank == 6
    assert torch.equal(oc.A, ref.Gi_iGij_Gj) and torch.equal(oc.B, ref.iGij) and torch.equal(oc.C, ref.Gj_iGij)
    assert abs(oc.fac - float(ref.fac)) < 1e-15


def test_time_grid_accumulates_like_the_reference():
    from semiclassical_amd import hostmath
    dt = 0.1
    t, ref = 0.0, []
    for _ in range(50):
        ref.append(t)
        t += dt
    assert np.array_equal(hostmath.time_grid(50, dt), np.array(ref))
    assert hostmath.time_grid(50, dt)[30] != 30 * dt        # the accumulated sum is not k*dt


def test_library_exports_every_declared_symbol():
    """every function declared in include/semiclassical_hip.h is exported by the shared library"""
    header = open(os.path.join(ROOT, "include", "semiclassical_hip.h")).read()
    declared = set(re.findall(r"\b(sc_[a-z_0-9]+)\s*\(", header))
    from semiclassical_amd import _lib
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    so = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(so, name), name
    assert _lib.lib.sc_version() >= 2
    assert _lib.lib.sc_abi_version() == _lib.ABI_VERSION
    assert int(re.search(r"#define SC_ABI_VERSION\s+(\d+)", header).group(1)) == _lib.ABI_VERSION
    for st in _lib.STRUCTS:
        assert _lib.lib.sc_struct_size(st.__name__.encode()) == ctypes.sizeof(st), st.__name__
    assert _lib.lib.sc_struct_size(b"no_such_struct") == -1
    # struct sizes agree with the C layout (LP64): a mismatch would shift every pointer
    assert ctypes.sizeof(_lib.sc_state) == 8 + 4 + 4 + 7 * 8
    assert ctypes.sizeof(_lib.sc_potential) == 4 + 4 + 3 * 8 + 8 + 8 + 8 + 8
    assert ctypes.sizeof(_lib.sc_hk_consts) == 16 + 6 * 8
    assert ctypes.sizeof(_lib.sc_overlap_consts) == 8 + 5 * 8 + 8
    assert ctypes.sizeof(_lib.sc_nac_consts) == 8 + 4 * 8 + 16


def test_product_library_reads_no_environment():
    """the product build has no experiment knobs: it does not even import getenv (those live behind -DSC_TUNING)"""
    import subprocess
    from semiclassical_amd import _lib
    assert _lib.lib.sc_tuning_build() == 0
    if os.environ.get("SC_LIB_PATH"):
        pytest.skip("a variant library is loaded")
    import shutil
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("nm not available")
    out = subprocess.run([nm, "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0 and "hipLaunchKernel" in out.stdout
    assert "getenv" not in out.stdout


def test_no_cpu_path():
    from semiclassical_amd import propagators as PR
    with pytest.raises(RuntimeError, match="no CPU path"):
        PR.HermanKlukPropagator(torch.eye(2), torch.eye(2), device="cpu")


def test_potential_torch_protocol_matches_oracle():
    """harmonic_approximation of the API classes (used outside the hot loop) == oracle potentials"""
    from semiclassical_amd import potentials as P
    from tests.engine_cases import engine_potential
    for name in ("hk_as5_chi002", "hk_as5_chi000", "hk_methylium", "hk_1d"):
        g = cases.load(name)
        pot, ref = engine_potential(g), cases.oracle_potential(g)
        r = torch.from_numpy(g["zi"][:pot.dimensions(), :7].copy())
        for a, b in zip(pot.harmonic_approximation(r), ref.harmonic_approximation(r)):
            assert torch.allclose(a, b, rtol=1e-14, atol=1e-300)
        assert torch.equal(pot.derivative_coupling_1st(r), ref.derivative_coupling_1st(r))


def test_fchk_reader_methylium():
    """zero-point energy, rank of Gamma_0 and adiabatic gap quoted in SURVEY.md section 8c"""
    ref_dir = "/root/reference/tests/DATA/examples/methylium_AH"
    if not os.path.isdir(ref_dir):
        pytest.skip("reference data not present on this machine")
    from semiclassical_amd import units
    from semiclassical_amd.readers import FormattedCheckpointFile
    with open(os.path.join(ref_dir, "opt_freq_s1.fchk")) as f:
        s1 = FormattedCheckpointFile(f)
    x0, G0, ezpt = s1.vibrational_groundstate()
    assert abs(ezpt - 0.0247324) < 1e-6
    assert np.linalg.matrix_rank(G0, tol=1e-8) == 6
    g = cases.load("hk_methylium")
    assert cases.rel_err(G0, g["Gamma_0"]) < 1e-12 and cases.rel_err(x0, g["q0"]) < 1e-15


def test_rk4_step_matrix_of_constant_hessian_equals_staged_rk4():
    """MolecularHarmonicPotential._step_matrix: Phi(dt) applied to random monodromy blocks = the oracle's four RK4
    stages of the same linear equations (reference propagators.py:86-119, 342-357)"""
    import numpy as np
    from semiclassical_amd import potentials as P
    rng = np.random.default_rng(2)
    D, dt = 7, 5.0
    masses = rng.uniform(1800, 20000, D)
    H = rng.standard_normal((D, D)) * 0.2
    H = H @ H.T
    pot = P.MolecularHarmonicPotential.from_arrays(np.zeros(D), np.float64(0.0), np.zeros(D), H, masses, np.zeros(D))
    phi = pot._step_matrix(dt)
    assert phi.shape == (2 * D, 2 * D) and P.MorsePotential._step_matrix(pot, dt) is None
    y = rng.standard_normal((2 * D, 2 * D))
    G = np.block([[np.zeros((D, D)), np.diag(1 / masses)], [-H, np.zeros((D, D))]])
    k1 = G @ y
    k2 = G @ (y + 0.5 * dt * k1)
    k3 = G @ (y + 0.5 * dt * k2)
    k4 = G @ (y + dt * k3)
    staged = y + dt / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    assert np.max(np.abs(phi @ y - staged)) < 1e-14 * np.max(np.abs(staged))
    big = P.MolecularHarmonicPotential.from_arrays(np.zeros(20), np.float64(0.0), np.zeros(20), np.eye(20),
                                                   np.ones(20), np.zeros(20))
    assert big._step_matrix(dt) is None            # the kernel taking Phi holds D <= 16


def test_dpp_hazards():
    """The inline-assembly DPP multiply-adds of csrc/sc_row16.h bypass the compiler's hazard recogniser: no DPP operand
    of the BUILT library may be read within two wait states of a VALU write of the same register."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "check_dpp_hazards", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                          "check_dpp_hazards.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    from semiclassical_amd import build
    if not os.path.exists(chk.OBJDUMP) or not os.path.exists(chk.BUNDLER):
        pytest.skip("ROCm LLVM tools not found")
    bad, n_dpp = chk.scan(chk.disassemble(build.LIB))[:2]
    assert n_dpp > 1000, "the disassembly did not reach the device code"
    assert not bad, bad[:5]


def test_cursor_handover_is_fenced_in_the_source():
    """hk_step_sd_kernel hands the next trajectory index from thread 0 to the other wavefronts through LDS; the round-2 GPU
    memory fault (docs/NOTEBOOK.md section 8.2) was a variant in which no barrier lay between the write and the reads.  The barrier
    must be unconditional (not inside a variant macro or a template branch) and sit between the two in the source."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "semiclassical_amd", "csrc", "sc_hk_step_sd.hip")).read()
    src = "\n".join(line.split("//")[0] for line in src.splitlines())            # comments stripped
    write = src.index("nextbuf[par] = drawn;")
    reads = [m.start() for m in re.finditer(r"readfirstlane\(nextbuf\[par\]\)", src)]
    assert reads and all(r > write for r in reads)
    code = src[write:min(reads)]
    assert "__syncthreads();" in code
    barrier_line = [l for l in code.splitlines() if "__syncthreads();" in l][0]
    assert barrier_line.strip() == "__syncthreads();", "the reset barrier must not be conditional"
    assert "#if" not in code[:code.index("__syncthreads();")] and "#endif" not in code[:code.index("__syncthreads();")]
    # the two LDS words are defined before the first trajectory
    assert re.search(r"nextbuf\[tid\] = 0; weakbuf\[tid\] = 0;", src)


def test_normal_mode_step_matrices_reproduce_the_full_step_matrix():
    """MolecularHarmonicPotential._normal_modes (sc_hk_run_modal): T blockdiag-by-mode(phi) T^-1 with T = diag(A, B) is the RK4 step
    matrix Phi(dt) of _step_matrix, including zero-frequency modes (free motion) and a wide mass ratio"""
    from semiclassical_amd import potentials as P
    rng = np.random.default_rng(7)
    D, dt = 12, 0.8
    masses = rng.uniform(1800.0, 2.2e4, D)
    W = rng.standard_normal((D, D - 6))
    H = (W @ W.T) * 0.05                                  # rank D - 6: six zero modes, as for a molecule in Cartesian coordinates
    H = H * np.sqrt(np.outer(masses, masses)) / 1.0e4
    pot = P.MolecularHarmonicPotential.from_arrays(np.zeros(D), np.float64(0.0), np.zeros(D), H, masses, np.zeros(D))
    A, B, Ainv, Binv, phi = pot._normal_modes(dt)
    assert np.max(np.abs(Ainv @ A - np.eye(D))) < 1e-12 and np.max(np.abs(Binv @ B - np.eye(D))) < 1e-12
    assert np.max(np.abs(B.T @ A - np.eye(D))) < 1e-12   # what makes G = Mp'^T Mq' invariant
    T = np.block([[A, np.zeros((D, D))], [np.zeros((D, D)), B]])
    Ti = np.block([[Ainv, np.zeros((D, D))], [np.zeros((D, D)), Binv]])
    mid = np.block([[np.diag(phi[:, 0]), np.diag(phi[:, 1])], [np.diag(phi[:, 2]), np.diag(phi[:, 3])]])
    full = pot._step_matrix(dt)
    assert np.max(np.abs(T @ mid @ Ti - full)) < 1e-12 * np.max(np.abs(full))
