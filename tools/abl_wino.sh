#!/bin/bash
# timing ablations of the Winograd kernel phases (results are wrong by construction): 1 transform, 2 MFMA, 4 filter->LDS, 8 patch->LDS
for a in 0 1 2 3 4 8 5 13 15; do echo "ablate=$a"; NO_LIB=1 DHZ_WINO_ABLATE=$a python tools/bench_wino.py 2>&1 | grep "C  512 K  512\|C   64 K   64" ; done
