import os, subprocess, sys
CASES = {
 "small_noeager": "st=bench.AttackStepper('RAFT',128,160,dev,3); st.enable_graph(); print(float(st.closure()))",
 "small_eager": "st=bench.AttackStepper('RAFT',128,160,dev,3); st.optimizer.zero_grad(); l=st._closure_body(); st.enable_graph(); print(float(st.closure()))",
 "small_step": "st=bench.AttackStepper('RAFT',128,160,dev,3); st.step(); st.enable_graph(); print(float(st.closure()))",
 "mid_noeager": "st=bench.AttackStepper('RAFT',256,320,dev,3); st.enable_graph(); print(float(st.closure()))",
 "small_closure_only": "st=bench.AttackStepper('RAFT',128,160,dev,3)\nfrom pcfa_amd.graphed import GraphedClosure\ng=GraphedClosure(st._closure_body,[st.nw1,st.nw2]); print(float(g()))",
 "small_fwd_only": "st=bench.AttackStepper('RAFT',128,160,dev,3)\nfrom pcfa_amd.graphed import GraphedForward\ng=GraphedForward(st._repredict_body, dev); print(g()[2].shape)",
}
pre = "import sys,torch; sys.path.insert(0,'.'); import bench; dev=torch.device('cuda')\n"
for name, code in CASES.items():
    r = subprocess.run([sys.executable, "-c", pre + code], capture_output=True, text=True, timeout=300)
    tail = (r.stdout.strip().splitlines() or [""])[-1]
    print("%-20s rc=%d  %s" % (name, r.returncode, tail[:80]), flush=True)
