"""BASELINE config #4 shape on the stand-in scene: 1920x1080, pure progressive photon mapping (numVplLightPaths = 0
disables the gather, rtcomphoton.h:200-203), 300 000 light paths, 100 iterations, through evplp_render_json."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import evplp_amd as ev
d = "/tmp/evplp_cfg4"
jp = ev.synth_scene(d, "conf", 331000, 1234, 1920, 1080)
over = dict(numLightPaths=300000, numVplLightPaths=0, radiusPercentage=0.003, DoProgressive=True, AlphaProgressive=0.7, numMaxIteration=int(sys.argv[1]) if len(sys.argv) > 1 else 100,
            timeLimitMs=1e9, frameMode="accumulate", run={"photonSplat": True}, combinedFilename="c.pfm", weightedVplFilename="v.pfm", weightedPhotonFilename="p.pfm", statFilename="stat.json", useStat=True)
t0 = time.time()
ev.render_json(jp, json.dumps(over))
print("wall incl. scene load + BVH build + image writes: %.2f s" % (time.time() - t0))
print(open(os.path.join(d, "stat.json")).read())
img = ev.load_pfm(os.path.join(d, "c.pfm"))
print("combined mean", img.mean(axis=(0, 1)))
os.makedirs("gpurun_out", exist_ok=True)
ev.save_image("gpurun_out/cfg4_ppm.png", img)
for k in ("p", "v"):
    im = ev.load_pfm(os.path.join(d, k + ".pfm")); print(k, "mean", im.mean(axis=(0, 1)), "max", im.max(), "nonzero frac", float((im.sum(-1) > 0).mean()))
