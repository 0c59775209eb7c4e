OUT=$PWD/gpurun_out; mkdir -p $OUT
run() {
  for C in 3 4; do
    HIKARI_HIP_LIB=$2 timeout 600 python bench.py --config $C --passes --no-cpu-baseline --no-hbm-probe --blocks 3 > $OUT/r04q_$1_config$C.json 2> $OUT/r04q_$1_config$C.err
    python - <<PY
import json
try:
    d = json.loads(open("$OUT/r04q_$1_config$C.json").read().strip().splitlines()[-1])
    print("$1 config $C:", d["ms_per_step"], "ms indirect alone", d["pass_ms"]["indirect_lit_ambient"], d["replay_bit_identical"])
except Exception as e:
    print("$1 failed", e, open("$OUT/r04q_$1_config$C.err").read()[-400:])
PY
  done
}
run base $PWD/bevy-hikari_amd/libhikari_hip.so
for v in w4s1 w4s3 w5s2 w6l16 w8l16 w3s2; do run $v $PWD/build_ab/$v.so; done
