"""Drop-in at the source level ("earth_moon.cc / saturn_ring.cc-shaped mains compile
unchanged"): the reference's OWN scene sources, unmodified and read from where they lie under
/root/reference (never copied), are compiled against THIS repository's host headers and -- for
the earth-moon scene, BASELINE config 1 -- run in full (400 big steps x 324 000 small steps) next
to the same source compiled against the reference's own header: the output files must be
identical byte for byte.  The recipe is `make -C oracle ref-mains` (outputs under oracle/_ref/,
which is never committed); the sources are fed to the compiler through stdin so that their
`#include "particle_simulation.h"` resolves through -I (a quoted include otherwise finds the
header next to the including file first).

Runs only where the reference is mounted (the build container); nothing here is marked gpu,
and nothing that runs on the GPU box reads /root/reference."""
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
REF_SRC = Path("/root/reference/src")
REFDIR = REPO / "oracle" / "_ref"
SCENES = ["earth_moon", "slingshoot", "earth_7moons", "saturn_ring"]

pytestmark = pytest.mark.skipif(not REF_SRC.exists(), reason="the reference is not mounted here")


@pytest.fixture(scope="module")
def built():
    if not (REPO / "n-body-sim_amd" / "lib" / "libnbody_mi355x.so").exists():
        pytest.fail("libnbody_mi355x.so not built: run __graft_entry__.build()")
    r = subprocess.run(["make", "-C", str(REPO / "oracle"), "ref-mains"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout + r.stderr


def test_reference_scene_sources_compile_unmodified_and_warning_free(built):
    """-Wall, the reference's documented flags (README.md:38): it builds warning-free with its
    own header, so it must with this one."""
    assert "warning" not in built, built
    for s in SCENES:
        assert (REFDIR / f"{s}_ourhdr").exists()


def test_earth_moon_full_run_is_byte_identical_to_the_reference(built, tmp_path):
    """BASELINE config 1, exactly as earth_moon.cc sets it up: 129.6 M small steps, SimulateCPU(),
    one thread, then WriteParticleFiles("")."""
    outs = {}
    for name in ("earth_moon_ourhdr", "earth_moon_refhdr"):
        d = tmp_path / name
        d.mkdir()
        r = subprocess.run([str(REFDIR / name)], cwd=d, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        assert "Average execution time for a big step" in r.stdout
        outs[name] = {f.name: f.read_bytes() for f in sorted(d.glob("N_Body_*.txt"))}
    assert sorted(outs["earth_moon_ourhdr"]) == ["N_Body_Particle_00000.txt", "N_Body_Particle_00001.txt"]
    assert outs["earth_moon_ourhdr"] == outs["earth_moon_refhdr"]
    assert outs["earth_moon_ourhdr"]["N_Body_Particle_00001.txt"].count(b"\n") == 401   # snapshot 0 + 400 big steps


def test_one_line_swap_to_simulate_mi355x_builds():
    """What a user changes to move a scene to the GPU: the `Simulate` line (cf. saturn_ring.cu:70).
    `make -C oracle ref-scenes`; the binary runs on the GPU box
    (test_host_mirror.py::test_reference_saturn_ring_main_on_the_gpu)."""
    assert (REF_SRC / "saturn_ring.cc").read_text().count("my_sim.SimulateAVX2();") == 1
    r = subprocess.run(["make", "-C", str(REPO / "oracle"), "ref-scenes"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "built _ref/saturn_ring_mi355x" in r.stdout, r.stdout + r.stderr
    assert "warning" not in r.stderr, r.stderr
    # ... and the reference's GPU scene (saturn_ring.cu; only its #include of the CUDA definitions
    # changes, its SimulateGPU(32) call stays) is built by the same target
    assert (REFDIR / "saturn_ring_cu_main").exists()
