# tools/lane_crossover.sh: where does one lane per trial (cgp_lane4.hpp, flags 4) overtake one wavefront per trial (flags 2)?  EKF and GH-3
# sigma-point filter, full outputs, T = 500 and T = 10000, by batch (cgp_api.hip: the limits handed to choose_wave)
for M in ekf ghf; do for T in 500 10000; do for B in 1024 2048 3072 4096 6144 8192 12288 16384 24576 32768; do
  if [ $((B * T)) -le 40000000 ]; then for f in 2 4; do python tools/crlb_probe.py $B $T $f full 4 $M 2>/dev/null; done; fi
done; done; done
