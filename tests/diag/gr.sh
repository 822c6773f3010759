#!/bin/bash
# gpurun helper: runs a diagnostic script with its output kept under gpurun_out/r4/<name>.log and shown.   bash tests/diag/gr.sh <script> <name>
mkdir -p gpurun_out/r4
bash "$1" > "gpurun_out/r4/$2.log" 2>&1
rc=$?
cat "gpurun_out/r4/$2.log"
exit $rc
