"""Render scenes on the GPU and compare with the CPU oracle and the golden PNGs.  Dev tool (the pytest
-m gpu suite does the same checks in a structured way).  Usage: python tools/gpu_parity.py [name-substring ...]"""
import glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np
from PIL import Image
from raytracer_amd import renderer as R
from raytracer_amd.harness import scene_yaml as Y
import oracle as O

names = [a for a in sys.argv[1:] if not a.startswith('-')]
size = None
for a in sys.argv[1:]:
    if a.startswith('--size='): size = int(a.split('=')[1])
G = os.path.join(ROOT, 'tests/golden')
for p in sorted(glob.glob(G + '/scenes/images/*/*.yml')) + sorted(glob.glob(G + '/scenes/animations/*/*.yml')):
    rel = p.split('scenes/')[1][:-4]
    if names and not any(n in rel for n in names): continue
    kw = dict(depth=4, supersampling_level=1)
    if size: kw.update(width=size, height=size)
    rs = Y.load_scene(p, **kw)
    fs = rs.builder.flatten()
    t0 = time.time(); gpu, st = R.render(fs, rs.camera, rs.options()); t1 = time.time()
    cpu, cst = O.render(fs, rs.camera, rs.options()); t2 = time.time()
    ropt = rs.options(); ropt.reference_order = 1
    rgpu, rst = R.render(fs, rs.camera, ropt)
    d = np.abs(gpu - cpu)
    gu, cu = O.quantize(gpu), O.quantize(cpu)
    du = np.abs(gu.astype(int) - cu.astype(int)).max(axis=2)
    msg = f'{rel:36s} f64 maxabs {d.max():.3e} exact {(d.max(axis=2)==0).mean()*100:8.4f}%  u8 equal {(du==0).mean()*100:8.4f}% >1LSB {(du>1).sum()}  gpu {st["ms_kernel"]:.1f}ms cpu {(t2-t1)*1e3:.0f}ms'
    msg += f'  reforder-vs-fast maxabs {np.abs(rgpu - gpu).max():.2e} ({rst["ms_kernel"]:.1f}ms)  simd {st["lane_steps"] / max(1, 64 * st["wave_steps"]):.2f}'
    st = rst
    rays = ('rays_primary','rays_shadow','rays_reflect','rays_refract','node_visits','tri_tests','prim_tests')
    same = all(st[k] == cst[k] for k in rays)
    msg += '  counters ' + ('same' if same else 'DIFF ' + str({k:(st[k],cst[k]) for k in rays if st[k]!=cst[k]}))
    gp = G + '/renders/test/' + rel.split('images/')[-1] + '.png'
    if os.path.exists(gp) and not size:
        ref = np.array(Image.open(gp).convert('RGB'))
        dg = np.abs(ref.astype(int) - gu.astype(int)).max(axis=2)
        msg += f'  golden equal {(dg==0).mean()*100:8.4f}% >1LSB {(dg>1).sum()}'
    print(msg, flush=True)
