# usage (GPU box): bash tools/exp/sweep_tiles.sh -> ms/step for tile-policy knobs (same box, interleaved baseline)
R=$GRAFT_REPO_ROOT
ONLY="--no-cpu-baseline --rotate-batches 0 --no-fg-capped --no-device-targets --no-direct-head-forward --no-fp32-mfma --pipeline-examples 0 --no-profile"
for t in ${SWEEP:-"" "big_min_tiles=128" "big_min_tiles=256" "big_min_tiles=512" "" "big_min_tiles=640" "small_rem_max=100" "small_rem_max=200" "" "small_whole_max=512" "small_whole_max=2048" "small_m_split=2" "small_m_split=3" "small_m_split=4" ""}; do
  python $R/bench.py $ONLY ${t:+--tune $t} 2>/dev/null > /tmp/b.json
  python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print('tune', '$t' or 'default', d['value'], d['ms_per_step'])"
done
