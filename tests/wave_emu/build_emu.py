"""TEST INFRASTRUCTURE ONLY: builds tests/wave_emu/libpetlion_emu.so = the product's device + host source compiled with g++
against the lock-step wave emulator (tests/wave_emu/hip/hip_runtime.h).  Used by the `-m "not gpu"` tests to exercise the
*device source* on a machine without a GPU.  The package never loads this library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
SRC = os.path.join(ROOT, "petlion.jl_amd", "csrc", "petlion_hip.hip")
OUT = os.path.join(HERE, "libpetlion_emu.so")


def build(force=False):
    deps = [SRC] + [os.path.join(os.path.dirname(SRC), f) for f in os.listdir(os.path.dirname(SRC)) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "hip", "hip_runtime.h"))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", HERE, SRC, "-o", OUT, "-Wno-unused-variable"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
