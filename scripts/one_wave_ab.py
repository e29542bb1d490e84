"""A/B of the one-wave general kernel between two builds of the library (CDPR_LIB), 16 384 and 65 536 x 8, steady / switching."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["libcdpr_hip.so"]
env0 = dict(os.environ, SCAN_B="16384,65536")
src = open(os.path.join(ROOT, "scripts", "gen_lean_scan.py")).read()
code = src[src.index("code = r'''") + len("code = r'''"):src.index("''' % ROOT")] % ROOT
for rep in range(2):
    for lib in libs:
        for label, env in (("one wave", {"CDPR_GEN_LEAN": "0", "CDPR_GEN_SPLIT": "0"}), ("lean    ", {"CDPR_GEN_LEAN": "1"})):
            subprocess.run([sys.executable, "-c", code], env=dict(env0, LABEL=f"{lib} {label}", CDPR_LIB=lib, **env))
