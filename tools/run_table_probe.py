"""tooling: how long does k_run_table take on a freshly uploaded 26 M-surfel map, and on the same map again?  (run under rocprofv3 --kernel-trace)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from maskfusion_amd import stress, synth

st = stress.stream(4)
mf = stress.make_context(0)
rgb, depth, mask = st.frame(0)
mf.processFrame(rgb, depth, mask=mask, classIDs=[0, 41, 42, 43, 44])
room = synth.dense_room_map(st.scene, int(0.8 * stress.surfel_capacity(stress.NUM_GSURFELS)), last_time=1.0, furniture_above=4)
for rep in range(2):
    t0 = time.perf_counter(); mf.getBackgroundModel().uploadMap(room); mf.sync(); t1 = time.perf_counter()
    print("upload + run table", rep, round(t1 - t0, 3), "s")
    for k in range(3):
        t0 = time.perf_counter(); mf.setParam("rebuildRunTable", 1); mf.sync(); print("  rebuild", k, round((time.perf_counter() - t0) * 1e3, 3), "ms")
mf.close()
