"""The 64-row attention core names accumulator registers a[64:255] (product build: a[56:255]) literally in inline asm and places its VALU work in the MFMA gaps by
hand; what hipcc does around those statements cannot be checked at run time alone.  tools/attn64_audit.py compiles the file the way
the build does (product and lab flags) and fails when the compiler (a) parks a value in an accumulator register the kernel owns, (b)
spills to scratch, or (c) reads an asm MFMA's result inside the hazard window (tools/attn64_hazards.py).  No GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_attn64_code_object_audit():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn64_audit.py"), "--lab"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count("0 WAR, 0 RAW") == 5, r.stdout[-4000:]
