"""A frame of the benchmark integrator that never leaves HBM (SURVEY.md N4 + N1/N2 on the device): the city cut into batching points
on the host, their BVH8s and SVDAGs built on the GPU and left there (pgpu_build_upload_bvh8_device, pgpu_build_svdags_device), then
per frame: camera rays (pgpu_camera_rays) -> [two-level trace (pgpu_trace) -> miss splat into the fixed-point film (pgpu_splat_misses)
-> Lambert bounce (pgpu_lambert_bounce)] x depth -> film resolve. What the reference does with PathIntegrator on top of its task graph
(path_integrator.cpp:100-158, sampler_integrator.cpp:131-163). Prints one JSON line.
Paths that ended are dropped after every bounce (pgpu_compact_paths; `slots`: they keep their place as rays that cannot hit anything).
Usage: python tools/frame_bench.py [image side, default 2048] [spp, default 2] [depth, default 4] [city blocks, default 24] [builder: resident | host] [compact | slots]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pandora_amd as pa

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 4
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 24
builder = sys.argv[5] if len(sys.argv) > 5 else "resident"
compact = (sys.argv[6] if len(sys.argv) > 6 else "compact") == "compact"
dev = torch.device("cuda", 0)
ctx = pa.accel.Context(0)
ctx.set_keep_shading(True)
mesh = pa.scenes.g_city(blocks, 18750)
t0 = time.perf_counter()
tl = pa.accel.build_two_level(mesh.pos, mesh.idx, 65536, svdag_res=64, svdag_device_ctx=ctx, resident_ctx=ctx if builder == "resident" else None)
tl.upload(ctx, 0 if builder == "resident" else ctx.upload_mesh(mesh.pos, mesh.idx))
scene_s = time.perf_counter() - t0
size = 20.0 * blocks
cam = pa.accel.make_camera((0.1 * size, 14.0, 0.1 * size), (0.6 * size, 0.0, 0.6 * size), (0, 1, 0), 60, 1.0)
n = side * side * spp
sky, albedo = (0.9, 1.0, 1.3), (0.7, 0.6, 0.5)
film = torch.zeros((side * side, 3), dtype=torch.int64, device=dev)
image = torch.zeros((side * side, 3), dtype=torch.float32, device=dev)


def frame(seed):
    film.zero_()
    torch.cuda.synchronize()  # (torch's stream and the context's meet here; everything below runs on the context's stream)
    rays, paths = ctx.camera_rays(cam, side, side, spp, seed, 0, n, dev)
    traced, out = 0, None
    for d in range(depth):
        out, _ = ctx.trace(rays, out=None if compact else out)
        traced += int(rays.shape[1])
        ctx.splat_misses(film, out["obj"], paths, sky)
        if d + 1 < depth:
            rays = ctx.lambert_bounce(None, rays, out, albedo, paths)
            if compact:
                rays, paths = ctx.compact_paths(rays, paths)
                if rays.shape[1] == 0:
                    break
    ctx.film_resolve(film, image)
    return traced


frame(1)
torch.cuda.synchronize()
frames = 3
t0 = time.perf_counter()
traced = sum(frame(2 + i) for i in range(frames))
torch.cuda.synchronize()
el = time.perf_counter() - t0
img = image.cpu().numpy()
print(json.dumps({"workload": "path-traced frame of the %d-triangle city on one GPU, everything between the camera and the resolved film in HBM" % mesh.n_tris,
                  "image": [side, side], "spp": spp, "depth": depth, "paths_per_frame": n, "ended_paths": "dropped after every bounce" if compact else "keep their slots", "rays_traced_per_frame": traced // frames, "batching_points": tl.n_leaves,
                  "bvh8_builder": builder, "scene_ready_s_after_generation": scene_s, "ms_per_frame": 1e3 * el / frames, "Mpaths_per_s": n * frames / el / 1e6,
                  "Mrays_traced_per_s": traced / el / 1e6,
                  "image_mean": [float(x) for x in img.mean(axis=0)], "image_finite": bool(np.isfinite(img).all())}))
ctx.close()
