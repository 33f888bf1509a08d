cd /root/repo/tools
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGSFM_LOOK_TIMING -DGSFM_BACK_TIMING $v -o /tmp/bcb bench_chol_batch.hip 2>/dev/null
  echo "## $v"; timeout 120 /tmp/bcb | sed -e 's/ per factorisation + solve (graph replay)  info 0  |Ax - b| sampled [0-9.e-]*  doubles of L, y, x differing from the fused form://' -e 's/^row workgroup of the one-launch-per-column form, matrix 1182 alone, //'
done
