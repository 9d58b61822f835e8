import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, torch
import oracle_binding as ob
pkg = ob.pkg
np.set_printoptions(linewidth=200, precision=17)
name, w, h, spp, seed = 'cornell', 32, 24, 2, 1
scene = pkg.Scene(); cam = scene.build_named(name, w, h)
view = scene.view()
# intersect KAT: camera rays from the oracle
ctx = pkg.Context(0); ctx.set_scene(scene)
rays = np.array([ob.oracle_camera_ray(cam, x, y, 1 + x + y * w) for y in range(h) for x in range(w)])
hits = ctx.intersect(rays)
ohits = np.array([ob.oracle_intersect(view, r) for r in rays])
print('intersect batch: max abs diff', np.abs(hits - ohits).max(), 'hit count', (hits[:, 0] > 0).sum(), (ohits[:, 0] > 0).sum())
g = ctx.rng_doubles(0, 1, 2000); o = ob.mt_unit_doubles(1, 2000)
print('rng kat seq first mismatch', np.argwhere(g != o)[:5].ravel().tolist())
for policy in (0, 1):
    params = pkg.default_params(width=w, height=h, samples_per_pixel=spp, seed=seed, rng_policy=policy)
    ref_rgb, ref_cnt, ref_words, _ = ob.oracle_render(view, cam, params, threads=2)
    rgb = torch.zeros((h, w, 3), dtype=torch.float64, device='cuda')
    cnt = torch.zeros((h, w), dtype=torch.int32, device='cuda')
    words = torch.zeros((spp, h, w), dtype=torch.int32, device='cuda')
    ctx.render(cam, params, rgb.data_ptr(), cnt.data_ptr(), words.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g = words.cpu().numpy().astype(np.uint32)
    bad = np.argwhere(g != ref_words)
    print('policy', policy, 'word mismatches', len(bad), 'of', g.size, 'first', bad[:3].tolist())
    if len(bad):
        k, y, x = bad[0]
        print('  gpu', g[k, y, max(0,x-2):x+3], 'ref', ref_words[k, y, max(0,x-2):x+3])
    r = rgb.cpu().numpy()
    print('  max rel err', np.max(np.abs(r - ref_rgb) / np.maximum(np.abs(ref_rgb), 1.0)), 'sum gpu', r.sum(), 'ref', ref_rgb.sum())
    print('  gpu words hist', np.unique(g, return_counts=True)[0][:10], ' ref', np.unique(ref_words, return_counts=True)[0][:10])
