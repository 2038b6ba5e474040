mkdir -p gpurun_out
for nb in 256 128 512; do
  SW_NB=$nb timeout -k 10 400 python tools/sweep_solver.py tools/sweep6.json > gpurun_out/sweep6_nb$nb.log 2>&1 || exit 1
done
