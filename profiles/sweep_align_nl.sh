#!/bin/bash
# sweep of align_kernel's reads-in-flight per wave (AK_NL) / start threshold (AK_START_MIN) builds and the sub-batch size
ROOT=$(cd $(dirname $0)/.. && pwd); OUT=$ROOT/gpurun_out/sweep_nl; mkdir -p $OUT
MONI_BENCH_SAVE_INDEX=1 timeout -k 10 500 python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu > $OUT/build.json 2> $OUT/build.log || exit 1
IFS=";" read -ra CFGS <<< "${SWEEP:-default 125000;default 250000;nl32_16 125000;nl32_16 250000}"
unset IFS
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  lib=$ROOT/moni_align_amd/csrc/libmoni_hip.so; [ $1 != default ] && lib=$ROOT/moni_align_amd/csrc/variants/libmoni_hip_$1.so
  export MONI_ALIGN_SUB=$2; [ "$2" = auto ] && unset MONI_ALIGN_SUB
  MONI_HIP_LIB=$lib MONI_AK_PROFILE=1 timeout -k 10 300 python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/b_$1_$2.json 2> $OUT/b_$1_$2.log || { echo "$cfg failed"; tail -3 $OUT/b_$1_$2.log; exit 1; }
  echo "$cfg: $(grep align_kernel $OUT/b_$1_$2.log | tail -1)"
  grep 'align_core wall' $OUT/b_$1_$2.log | tail -2
  python3 -c "import json;d=json.loads(open('$OUT/b_$1_$2.json').read().strip().splitlines()[-1]);print('   ', round(d['value']), d['ms_per_step'], d['stages_s_per_step'])"
done
