"""Dev tool (CPU, numpy): how many of the 26 candidates of a voxel can be skipped for free in each JFA pass -- same seed as the own state (equal distance never wins,
sequential.cpp:106) or no seed -- and how often that holds for a whole 64-wide run of x (what a wave-uniform branch could skip).  bunny, n = 128.
  python tools/skippable_candidates.py"""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import oracle as O
from cuda_mesh_voxelization_amd import mesh as M
from cuda_mesh_voxelization_amd.capi import Frame
from slab_cpu_backend import CpuSlabBackend, NONE
import torch
n=128
xyz,tri=M.import_mesh(M.asset('bunny.obj')); o,vs=O.frame([xyz],n); fr=Frame.make(n,vs,o)
be=CpuSlabBackend((xyz,tri))
words=be.empty_u32(fr.words); be.voxelize(fr,words,None,None,2)
ids=torch.zeros(fr.voxels,dtype=torch.int32); be.jfa_init(fr,words,None,None,ids)
A=ids.numpy().reshape(n,n,n).copy()
k=n//2
while k>=1:
    none_plane=np.full((n,n),NONE,np.int32)
    def plane(z,d,A=A,k=k):
        zz=z+d*k
        return A[zz] if 0<=zz<n else none_plane
    # stats on input state A for this pass
    tot=0; eq=0; runs=0; runs_eq=0; W=64
    for dz in (-1,0,1):
        for dy in (-1,0,1):
            for dx in (-1,0,1):
                if dx==dy==dz==0: continue
                B=np.full_like(A,NONE)
                zs=slice(max(0,-dz*k),n-max(0,dz*k)); ys=slice(max(0,-dy*k),n-max(0,dy*k)); xs=slice(max(0,-dx*k),n-max(0,dx*k))
                zt=slice(max(0,dz*k),n-max(0,-dz*k)); yt=slice(max(0,dy*k),n-max(0,-dy*k)); xt=slice(max(0,dx*k),n-max(0,-dx*k))
                B[zs,ys,xs]=A[zt,yt,xt]
                e=(B==A)|(B==NONE)          # candidate cannot win: same seed as own, or none
                tot+=e.size; eq+=int(e.sum())
                r=e.reshape(n,n,n//W,W).all(axis=3)
                runs+=r.size; runs_eq+=int(r.sum())
    print("k=%3d  candidates that cannot win (same id / none): %.3f   64-wide x-runs where the whole direction is skippable: %.3f  seeds: %.3f" % (k, eq/tot, runs_eq/runs, float((A!=NONE).mean())))
    A=be._pass(fr,k,plane)
    k//=2
