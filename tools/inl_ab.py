import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import rrt_amd  # noqa
from radiance_ray_tracing_amd import rd, scenes
plt = rd.Platform.GetPlatform(0)
dev = scenes.DeviceScene(scenes.CONFIGS["c1_cornell"](1920, 1080, 4, 8), plt)
for inl in (1, 0, 1, 0):
    for cull in (-1, 1):
        rd.SetOption("inline_leaf_roots", inl); rd.SetOption("cull", cull)
        for _ in range(3):
            dev.set_rtprop(totalSamples=0); rd.TraceRays(plt, 0, 0, 0, 1920, 1080)
        t0 = time.perf_counter()
        for _ in range(10):
            dev.set_rtprop(totalSamples=0); rd.TraceRays(plt, 0, 0, 0, 1920, 1080)
        print("inline_leaf_roots", inl, "cull", cull, "%.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
