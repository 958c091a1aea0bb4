#!/bin/bash
# round 4, GPU call H: issue-rate probe (tools/probe/pipe_probe.hip): v_exp_f32 / softmax VALU mix / MFMA alone and interleaved in one wave.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04_h; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/pipe_probe tools/probe/pipe_probe.hip > $O/build.log 2>&1
timeout 120 /tmp/pipe_probe > $O/pipe_probe.json 2> $O/pipe_probe.err; echo "rc $?"
cat $O/pipe_probe.json
