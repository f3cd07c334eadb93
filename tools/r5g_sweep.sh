# usage (GPU box): bash tools/r5g_sweep.sh  -- times the 16x16 raster launch group for a few group-kernel configurations (env switches are read once per process)
for cfg in "VVCGPU_NO_R5GQ=1" "VVCGPU_R5G_NB=8" "VVCGPU_R5G_NB=8 VVCGPU_R5G_RPS=6" "VVCGPU_R5G_NB=8 VVCGPU_R5G_RPS=9" "VVCGPU_R5G_NB=8 VVCGPU_R5G_RPS=6 VVCGPU_R5G_KB=30" "VVCGPU_R5G_NB=8 VVCGPU_R5G_RPS=3"; do
  echo "== $cfg"
  env $cfg python3 tools/run_stage.py --only 16x16_39 --reps 8 2>/dev/null
done
