for t in 0 1 2; do echo "PCFA_SC_TILE=$t"; PCFA_SC_TILE=$t python tools/bench_scorr.py kitti 100 2>/dev/null | grep "device time"; done
for n in 1 2 4; do echo "PCFA_SC_TILE=1 NS=$n"; PCFA_SC_TILE=1 PCFA_SC_NS=$n python tools/bench_scorr.py kitti 100 2>/dev/null | grep "device time"; done
echo "BTH=4"; PCFA_SC_BTH=4 python tools/bench_scorr.py kitti 100 2>/dev/null | grep "device time"
tools/run_matrix_box.sh GMA "6,7" 20 16 > /dev/null 2>&1; echo matrix rc=$?
