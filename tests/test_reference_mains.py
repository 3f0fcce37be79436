"""Drop-in at the source level ("earth_moon.cc / saturn_ring.cc-shaped mains compile
unchanged"): the reference's OWN scene sources, unmodified and read from where they lie under
/root/reference (never copied), are compiled against THIS repository's host headers and -- for
the earth-moon scene, BASELINE config 1 -- run in full (400 big steps x 324 000 small steps) next
to the same source compiled against the reference's own header: the output files must be
identical byte for byte.  The sources are fed through stdin so that their `#include
"particle_simulation.h"` resolves through -I (a quoted include otherwise finds the header next
to the including file first).

Runs only where the reference is mounted (the build container); nothing here is marked gpu,
and nothing that runs on the GPU box reads /root/reference."""
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
REF_SRC = Path("/root/reference/src")
HOST = REPO / "n-body-sim_amd" / "host"
LIBDIR = REPO / "n-body-sim_amd" / "lib"
FLAGS = ["-std=c++17", "-Wall", "-O3", "-mavx2", "-mfma", "-pthread"]   # reference README.md:38 (+ -mfma, as the oracle)

pytestmark = pytest.mark.skipif(not REF_SRC.exists(), reason="the reference is not mounted here")


def compile_against_ours(source_text, out):
    if not (LIBDIR / "libnbody_mi355x.so").exists():
        pytest.fail("libnbody_mi355x.so not built: run __graft_entry__.build()")
    r = subprocess.run(["g++", *FLAGS, f"-I{HOST}", f"-I{REPO / 'include'}", "-x", "c++", "-", "-o", str(out),
                        f"-L{LIBDIR}", "-lnbody_mi355x", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib"],
                       input=source_text, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "warning" not in r.stderr, r.stderr   # the reference builds warning-free with -Wall; so must this
    return out


@pytest.mark.parametrize("scene", ["earth_moon", "slingshoot", "earth_7moons", "saturn_ring"])
def test_reference_scene_sources_compile_unmodified(tmp_path, scene):
    compile_against_ours((REF_SRC / f"{scene}.cc").read_text(), tmp_path / scene)


def test_one_line_swap_to_simulate_mi355x_compiles(tmp_path):
    """What a user changes to move a scene to the GPU: the `Simulate` line (cf. saturn_ring.cu:70)."""
    src = (REF_SRC / "saturn_ring.cc").read_text()
    assert src.count("my_sim.SimulateAVX2();") == 1
    compile_against_ours(src.replace("my_sim.SimulateAVX2();", "std::cout << my_sim.SimulateMI355X() << std::endl;"),
                         tmp_path / "saturn_ring_mi355x")


def test_earth_moon_full_run_is_byte_identical_to_the_reference(tmp_path):
    """BASELINE config 1, exactly as earth_moon.cc sets it up: 129.6 M small steps, SimulateCPU(), one
    thread, then WriteParticleFiles("")."""
    src = REF_SRC / "earth_moon.cc"
    ours = compile_against_ours(src.read_text(), tmp_path / "ours")
    theirs = tmp_path / "theirs"
    r = subprocess.run(["g++", *FLAGS, str(src), "-o", str(theirs)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    outs = {}
    for name, exe in (("ours", ours), ("theirs", theirs)):
        d = tmp_path / f"run_{name}"
        d.mkdir()
        r = subprocess.run([str(exe)], cwd=d, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        assert "Average execution time for a big step" in r.stdout
        outs[name] = {f.name: f.read_bytes() for f in sorted(d.glob("N_Body_*.txt"))}
    assert sorted(outs["ours"]) == ["N_Body_Particle_00000.txt", "N_Body_Particle_00001.txt"]
    assert outs["ours"] == outs["theirs"]
    assert outs["ours"]["N_Body_Particle_00001.txt"].count(b"\n") == 401   # snapshot 0 + 400 big steps
