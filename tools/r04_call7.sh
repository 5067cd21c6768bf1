for v in base "" nogelu nogelunop2 nodma nodmanobar; do
  L=""; [ -n "$v" ] && L=$PWD/facialmmt_amd/libfmmt_hip_$v.so
  PROBE_LIB=$L timeout 200 python tools/probes/mlp_fwd_only.py 2>&1 | grep -v amdgpu.ids
done
