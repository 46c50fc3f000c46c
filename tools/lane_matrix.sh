# tools/lane_matrix.sh [T...]: the large-batch EKF by outputs wanted (mfs, Pfs, nll) and record length, lane4 kernel (flags 4) against the round-4 lane kernel (0x14)
for T in ${@:-500 512}; do for w in 111 110 100 101; do for f in 4 0x14; do python tools/crlb_probe.py 262144 $T $f $w 4 2>/dev/null; done; done; done
