#!/bin/bash
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r3zn; O=$R/gpurun_out/r3zn; cd $R
timeout 600 python examples/dipole_sphere_absorber.py > $O/example.log 2> $O/example.err; cat $O/example.log; tail -3 $O/example.err
