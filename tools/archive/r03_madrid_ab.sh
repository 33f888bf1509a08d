# Madrid (three configurations of tools/madrid_time.py) under the backward-substitution variants of run_dense
for v in "GSFM_CHOL_BACK_GROUP=16" "GSFM_CHOL_BACK_GROUP=8" "GSFM_CHOL_BACK_GROUPS=0" "GSFM_CHOL_BACK_GROUP=16" "GSFM_CHOL_BACK_GROUP=8"; do echo "$v"; env $v timeout 200 python tools/madrid_time.py 2>&1 | grep "^Madrid" | head -3; done
