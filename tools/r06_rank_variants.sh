# round 6: one rank's share of the C5 graph (tools/r03_rank_share_probe.py) under the single-round kernel variants, one box, one after the other.
# K3c: sub-chunks per iteration (GSFM_K3C_EPL); K2c: lanes per workgroup (GSFM_K2C_THREADS); K1 direct: trips per chunk (GSFM_K1D_TRIPS) and rolling
# requests (a second library built with -DGSFM_K1D_ROLL=0: tools/_ab/libgsfm_rot_noroll.so).
cd "$(dirname "$0")/.."
W=${1:-8}
run() { echo "## $*"; env "$@" timeout 200 python tools/r03_rank_share_probe.py $W 2>&1 | grep -v "^GSFM_COL_WGS=default $" | cut -c1-260; }
run GSFM_K3C_EPL=1 GSFM_K2C_THREADS=256 GSFM_K1D_TRIPS=3 GSFM_ROT_LIB=$PWD/tools/_ab/libgsfm_rot_noroll.so
run GSFM_K3C_EPL=2 GSFM_K2C_THREADS=512 GSFM_K1D_TRIPS=3
run GSFM_K3C_EPL=2 GSFM_K2C_THREADS=512
run GSFM_K3C_EPL=2 GSFM_K2C_THREADS=512 GSFM_K1D_TRIPS=2
run GSFM_K3C_EPL=1 GSFM_K2C_THREADS=256 GSFM_K1D_TRIPS=8
