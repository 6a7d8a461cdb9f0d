#!/usr/bin/env python3
"""Would a float32 first pass pay?  CPU-only analysis on the benchmark's own pairs (40 of the vc workload's 1 000): the
composed projection evaluated in float32 (FMAs emulated exactly) against float64 -- the error in the projected pixel and in the
camera-2 depth -- and, for three guard widths, the share of in-view lanes that would have to be re-evaluated in float64 and
the share of 64-lane rows that contain such a lane (what decides between a row-level and a lane-level second pass).
    python tools/fp32_pass_stats.py > profiles/r03_fp32_pass_stats.md
"""
import os, sys
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0]=[ROOT+'/multi-spatialmllm_amd', ROOT, ROOT+'/tools']
import bench
from mspa import engine, workload
from oracle import np_oracle as O
H,W=480,640
sys.argv=[sys.argv[0]]
args=bench.parse_args()
sc=bench.make_base_scene(args,0)
ids=sc.valid_image_ids
masks=O.scene_visibility_masks(sc.points[:,:3], sc.K, sc.A, sc.E, sc.depth, (H,W))
n=len(ids)
ov=np.array([O.calculate_camera_overlap(masks[ids[i]],masks[ids[j]]) for i in range(n) for j in range(i+1,n)])
pairs,info=workload.select_pairs(ov,n,1000,"vc",seed=77)
mats=engine.frame_matrices(sc.K,sc.A,[sc.E[i] for i in ids])
rng=np.random.default_rng(0)
sel=rng.choice(len(pairs),40,replace=False)
yy,xx=np.mgrid[0:H,0:W]
f32=np.float32
def fma32(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
errs_u=[]; errs_z=[]; tot=0
guards=[(1e-3,0.01),(2e-3,0.02),(5e-3,0.05)]
cnt={g:0 for g in guards}; rows_any={g:0 for g in guards}; nrows=0
for p in sel:
    a,b=pairs[p]
    M=(mats[b,6].reshape(4,4)@mats[a,5].reshape(4,4))[:3].copy()
    M[:,3]*=1000.0
    d=sc.depth[ids[a]].astype(np.float64)
    # fp64 composed
    t=M[:,0][None,None,:]*xx[...,None]+M[:,1][None,None,:]*yy[...,None]+M[:,2][None,None,:]
    q=t*d[...,None]+M[:,3]
    with np.errstate(all='ignore'):
        u=q[...,0]/q[...,2]; v=q[...,1]/q[...,2]
    iz=q[...,2]
    # fp32
    M32=M.astype(f32); x32=xx.astype(f32); y32=yy.astype(f32); d32=d.astype(f32)
    t32=[fma32(np.full_like(x32,M32[k,1]),y32,fma32(np.full_like(x32,M32[k,0]),x32,np.full_like(x32,M32[k,2]))) for k in range(3)]
    q32=[fma32(t32[k],d32,np.full_like(d32,M32[k,3])) for k in range(3)]
    with np.errstate(all='ignore'):
        r=(f32(1)/q32[2]).astype(f32)
        u32=(q32[0]*r).astype(f32); v32=(q32[1]*r).astype(f32)
    inv=(d>0)&(u>-1)&(u<W+1)&(v>-1)&(v<H+1)&(iz>1)
    eu=np.maximum(np.abs(u32-u),np.abs(v32-v))[inv]; ez=np.abs(q32[2]-iz)[inv]
    errs_u.append(eu); errs_z.append(ez)
    xi=np.clip(np.rint(np.nan_to_num(u)),0,W-1).astype(int); yi=np.clip(np.rint(np.nan_to_num(v)),0,H-1).astype(int)
    dv=sc.depth[ids[b]].astype(np.float64)[yi,xi]
    for g in guards:
        gp,gz=g
        fu=np.abs(u-np.rint(u)); fv=np.abs(v-np.rint(v))
        risky=inv&((fu<gp)|(fu>0.5-gp)|(fv<gp)|(fv>0.5-gp)|(np.abs(iz-dv)<gz))
        cnt[g]+=int(risky.sum())
        rows_any[g]+=int((risky.reshape(H,W//64,64).any(-1)).sum())
    tot+=int(inv.sum()); nrows+=int(inv.reshape(H,W//64,64).any(-1).sum())
eu=np.concatenate(errs_u); ez=np.concatenate(errs_z)
print('# float32 first pass: error and guarded share (tools/fp32_pass_stats.py, CPU analysis, 40 pairs of the vc workload)\n')
print('in-view lanes',tot, '\n')
print('* fp32 |du| px: median %.2e  99%% %.2e  99.99%% %.2e  max %.2e'%(np.median(eu),np.percentile(eu,99),np.percentile(eu,99.99),eu.max()))
print('* fp32 |dz| mm: median %.2e  99%% %.2e  99.99%% %.2e  max %.2e'%(np.median(ez),np.percentile(ez,99),np.percentile(ez,99.99),ez.max()))
for g in guards:
    print('* guard px %.0e / mm %.2f: guarded lanes %.3f %% of in-view, 64-lane rows with a guarded lane %.1f %%'%(g[0],g[1],100*cnt[g]/tot,100*rows_any[g]/nrows))

# ---- a guard that is a BOUND, not an observation: per tile B_k = c eps (Tmax_k dhi + |M_k3|) with Tmax_k = sum of the absolute
# terms of the affine row at the tile's far corner (no credit for cancellation), per lane g_u = (B_0 + |u| B_2) / |q_2|, likewise
# g_v; g_z = B_2 -- exactly the form of DESIGN.md section 8.1 (constants 4.02, 1.25, 3.01, + the float64 level's 1e-6).
eps = 2.0 ** -24
c = 4.02
viol = 0
lanes = 0
guarded = 0
rows_g = 0
rows_n = 0
gmax = 0.0
for p in sel:
    a, b = pairs[p]
    M = (mats[b, 6].reshape(4, 4) @ mats[a, 5].reshape(4, 4))[:3].copy()
    M[:, 3] *= 1000.0
    d = sc.depth[ids[a]].astype(np.float64)
    t = M[:, 0][None, None, :] * xx[..., None] + M[:, 1][None, None, :] * yy[..., None] + M[:, 2][None, None, :]
    q = t * d[..., None] + M[:, 3]
    with np.errstate(all='ignore'):
        u = q[..., 0] / q[..., 2]
        v = q[..., 1] / q[..., 2]
    iz = q[..., 2]
    M32 = M.astype(f32); x32 = xx.astype(f32); y32 = yy.astype(f32); d32 = d.astype(f32)
    t32 = [fma32(np.full_like(x32, M32[k, 1]), y32, fma32(np.full_like(x32, M32[k, 0]), x32, np.full_like(x32, M32[k, 2]))) for k in range(3)]
    q32 = [fma32(t32[k], d32, np.full_like(d32, M32[k, 3])) for k in range(3)]
    with np.errstate(all='ignore'):
        r = (f32(1) / q32[2]).astype(f32)
        u32 = (q32[0] * r).astype(f32).astype(np.float64)
        v32 = (q32[1] * r).astype(f32).astype(np.float64)
    z32 = q32[2].astype(np.float64)
    xi = np.clip(np.rint(np.nan_to_num(u)), 0, W - 1).astype(int)
    yi = np.clip(np.rint(np.nan_to_num(v)), 0, H - 1).astype(int)
    dv = sc.depth[ids[b]].astype(np.float64)[yi, xi]
    for R0 in range(0, H, 48):
        for c0 in range(0, W, 64):
            sl = (slice(R0, R0 + 48), slice(c0, c0 + 64))
            dt = d[sl]
            if not (dt > 0).any():
                continue
            dhi = dt.max()
            Tmax = np.abs(M[:, 0]) * (c0 + 63) + np.abs(M[:, 1]) * (R0 + 47) + np.abs(M[:, 2])
            B = c * eps * (Tmax * dhi + np.abs(M[:, 3]))
            inv = (dt > 0) & (u[sl] > -1) & (u[sl] < W + 1) & (v[sl] > -1) & (v[sl] < H + 1) & (iz[sl] > 1)
            if not inv.any():
                continue
            with np.errstate(all='ignore'):
                gu = 1.25 * ((B[0] + np.abs(u32[sl]) * B[2]) / np.abs(z32[sl]) + 3.01 * eps * np.abs(u32[sl])) + 1e-6
                gv = 1.25 * ((B[1] + np.abs(v32[sl]) * B[2]) / np.abs(z32[sl]) + 3.01 * eps * np.abs(v32[sl])) + 1e-6
            gz = B[2] + 1e-6
            viol += int((inv & ((np.abs(u32[sl] - u[sl]) > gu) | (np.abs(v32[sl] - v[sl]) > gv) | (np.abs(z32[sl] - iz[sl]) > gz))).sum())
            fu = np.abs(u32[sl] - np.rint(u32[sl])); fv = np.abs(v32[sl] - np.rint(v32[sl]))
            risky = inv & ((fu < gu) | (fu > 0.5 - gu) | (fv < gv) | (fv > 0.5 - gv) | (np.abs(z32[sl] - dv[sl]) < gz) | (np.abs(z32[sl]) <= 2 * B[2]))
            guarded += int(risky.sum()); lanes += int(inv.sum())
            rows_g += int(risky.any(1).sum()); rows_n += int(inv.any(1).sum())
            gmax = max(gmax, float(np.max(gu[inv])), float(np.max(gv[inv])))
print('\n* bound-derived guard (DESIGN.md 8.1: per tile B_k = 4.02 eps (Tmax_k dhi + |M_k3|), per lane 1.25 ((B_0 + |u| B_2) / |q_2| + 3.01 eps |u|) + 1e-6): lanes whose float32 result '
      'is off by MORE than their guard: %d of %d; guarded lanes %.3f %%; 64-lane rows with a guarded lane %.1f %%; largest guard %.2e px'
      % (viol, lanes, 100 * guarded / lanes, 100 * rows_g / rows_n, gmax))
