#!/bin/bash
# r04l: E stores of a plane issued one H phase later, behind the loads of the next plane (instantiations without CPML)
cd /root/repo; mkdir -p gpurun_out
: > gpurun_out/probe_r04l.jsonl
for w in v0 v1; do timeout 600 python scripts/probe_ab_held.py 512 $w "OPT_MEM_HINTS=1;OPT_MEM_HINTS=265" 3 >> gpurun_out/probe_r04l.jsonl 2>> gpurun_out/probe_r04l.err; done
grep "^{" gpurun_out/probe_r04l.jsonl
