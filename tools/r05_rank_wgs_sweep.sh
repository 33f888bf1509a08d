cd /root/repo
for spec in "8 500" "8 525" "8 275" "8 750" "4 490" "4 980" "4 735" "2 980" "2 1078" "2 490"; do
  set -- $spec
  GSFM_COL_WGS=$2 timeout 120 python tools/r03_rank_share_probe.py $1 2>&1 | grep "ranks $1" | cut -c1-220
done
