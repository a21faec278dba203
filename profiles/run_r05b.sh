#!/bin/bash
# round 4: 2-rank rehearsal on the one GPU (gloo; both ranks on device 0) of `bench.py --gpus 2` with two contexts per rank: sharding, chunk ownership, gather, verification against the unsharded text
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r05b; mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 400 python3 bench.py --steps 1 --warmup 0 --no-cpu --no-from-host --no-scaling-base --no-single-context > $OUT/build.json 2> $OUT/build.log || exit 1
reh() { local name=$1; shift
MONI_BENCH_BACKEND=gloo MONI_BENCH_DEVICE=0 timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu --verify-gather "$@" > $OUT/rehearse_$name.json 2> $OUT/rehearse_$name.log || { grep -v amdgpu.ids $OUT/rehearse_$name.log | tail -20; return 1; }
grep '"metric"' $OUT/rehearse_$name.json | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', {k: d[k] for k in ('value', 'n_gpus', 'ms_per_step', 'scaling', 'aligned_all_ranks')})
print('  ', d['config']['parallelism'], d['config']['chunks_per_rank'], 'contexts per rank', d['config']['contexts_in_flight'])
print('   gather', {k: d['gather'][k] for k in ('seconds', 'bytes', 'records_match_reads', 'identical_to_unsharded', 'backend')})
"
}
# (two ranks SHARE the one GPU here: a context's working memory for chunks of 1 M reads is ~45 GB, so the two-context form is rehearsed with chunks of 250 k reads)
reh inflight1 --total-reads 4000000 --inflight 1 && reh inflight2 --total-reads 2000000 --reads 250000 --inflight 2
