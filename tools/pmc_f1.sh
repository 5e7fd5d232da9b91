#!/bin/bash
# usage (GPU box): tools/pmc_f1.sh  -- SQ counter passes of tools/exp_f1_lattice.py, summary for the f1 kernels
KFILTER="consensus,lattice_count" exec $GRAFT_REPO_ROOT/tools/pmc_cmd.sh pmc_f1 tools/exp_f1_lattice.py 1
