"""Sweep of the limit-sharing schedule / split count of the coarse matcher on the KT shape (through umereg_match_opts, per call)."""
import os, sys, subprocess
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for mask in ("0x8000808b", "0x80008009", "0x80000081", "0x80000001", "0x80000889"):
    for sp in ("17", "10"):
        env = dict(os.environ, TUNE_SHARE_MASK=mask, TUNE_SPLITS=sp)   # read by exp_f16r_stats.py -> ops.MatchOpts
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_f16r_stats.py")], env=env, capture_output=True, text=True)
        print(mask, "splits", sp, out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
