#!/bin/bash
for v in lib_trace.so lib_trace_1.so lib_trace_2.so lib_trace_3.so lib_trace_7.so lib_trace_8.so lib_trace_15.so; do
  echo "== $v"; python tools/show_trace_arsb.py $v 2>&1 | grep "wave 0"
done
