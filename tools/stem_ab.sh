# GPU box: stem kernel experiments (U-Net/resnet50 704^2 step + the stem's own forward / weight-gradient time)
for e in "$@"; do
  ee="$e"; [ "$e" = "-" ] && ee="OCTSEG_NOP=1"
  env $ee OCTSEG_PROFILE_DUMP=gpurun_out/stem_l.csv python3 bench.py --workload unet_r50_704 --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['value'], d['ms_per_step'])"
  python3 tools/layer_table.py gpurun_out/stem_l.csv encoder.conv1 | grep "encoder.conv1 "
done
