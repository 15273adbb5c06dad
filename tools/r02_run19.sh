#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/pytest19.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest19.log
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product_new.so; bash tools/ab_libs.sh new /tmp/lib_product_new.so old moephoto_amd/_abl/lib_sp_old.so
