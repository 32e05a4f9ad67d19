set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
python scripts/fused_probe.py --split > gpurun_out/s10/base.log 2>&1
PYGAMD_EXTRA_HIPCC_FLAGS=-DPYGAMD_GATHER_NT=1 python scripts/fused_probe.py --split > gpurun_out/s10/nt.log 2>&1
python scripts/fused_probe.py --split > gpurun_out/s10/base2.log 2>&1
grep -v "amdgpu.ids\|pyg_amd build" gpurun_out/s10/base.log | tail -12
echo ---- NT
grep -v "amdgpu.ids\|pyg_amd build" gpurun_out/s10/nt.log | tail -12
echo ---- base again
grep -v "amdgpu.ids\|pyg_amd build" gpurun_out/s10/base2.log | tail -12
