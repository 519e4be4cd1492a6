"""CPU check of the word-at-a-time UTF-8 tests used by the decode kernels (wordpiece_amd/csrc/utf8_swar.h)
against the oracle's sequential decoder (utf8.cpp:54-90,130-147 restated): compiled with g++, no GPU."""
import os
import subprocess

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_swar_utf8_equals_sequential_decoder(tmp_path):
    O.lib()  # builds oracle/liboracle.so if needed
    exe = str(tmp_path / "test_utf8_swar")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_utf8_swar.cpp"),
                    "-L" + O.ODIR, "-loracle", "-Wl,-rpath," + O.ODIR, "-fopenmp"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok" in r.stdout
