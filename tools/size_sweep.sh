for n in 32 64 96 128 160 192 224 256 288 320 352 384 416 448 480 512 544 576 608 640 704 768 832 896 960 1024; do
  st=$(( 40000 / n )); [ $st -gt 300 ] && st=300; [ $st -lt 12 ] && st=12
  r=$(python tools/ab_wall.py --n $n --steps $st --rounds 3 --refine 1 --libs cuda_mesh_voxelization_amd/libvphip.so 2>/dev/null | grep median | awk '{print $3}')
  python -c "n=$n; t=$r; print('n=%4d  %.4f ms  %.1f ps/voxel  %.1f Gvox/s' % (n, t, t*1e9/n**3, n**3/t/1e6))"
done
