#!/bin/bash
# round 6, call ZF: counters of tail3 on a dn_lite5 frame and of the SEDN helper chain on an l25 frame
set -u
cd "$(dirname "$0")/.."
{
PMC_TAG=r06zf_a PMC_MODEL="DN lite5" PMC_RE="tail3|stem_kernel" bash tools/pmc_kernels.sh
PMC_TAG=r06zf_b PMC_MODEL="DN l25" PMC_RE="sedn_weff|sedn_fmean|sedn_xsum" bash tools/pmc_kernels.sh
} > gpurun_out/r06zf.txt 2>&1
cat gpurun_out/r06zf.txt
