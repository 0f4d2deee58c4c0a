"""Compile oracle/ssm_oracle.c -> oracle/_build/liboracle.so with gcc (+OpenMP). Building the checker is not using it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ssm_oracle.c")
OUT = os.path.join(HERE, "_build", "liboracle.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", OUT, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
