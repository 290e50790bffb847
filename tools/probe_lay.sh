mkdir -p gpurun_out/r05u
timeout 1500 python tools/mesh_probe.py motorbike:mb12 "lay:PROBE_SWEEPS=2" "nolay:PROBE_SWEEPS=2,LDU_GS_LAYOUTS=0" "lay_bpc2:PROBE_SWEEPS=2,LDU_P2P_BPC=2" "lay_bpc4:PROBE_SWEEPS=2,LDU_P2P_BPC=4" "lay_win0:PROBE_SWEEPS=2,LDU_P2P_WINDOW=0" > gpurun_out/r05u/probe_l0.log 2>&1
timeout 1500 python tools/mesh_probe.py motorbike:mb12@1,2 "lay:PROBE_SWEEPS=3,LDU_BLK=0" "nolay:PROBE_SWEEPS=3,LDU_GS_LAYOUTS=0,LDU_BLK=0" "lay_bpc2:PROBE_SWEEPS=3,LDU_P2P_BPC=2,LDU_BLK=0" "lay_bpc4:PROBE_SWEEPS=3,LDU_P2P_BPC=4,LDU_BLK=0" "lay_win0:PROBE_SWEEPS=3,LDU_P2P_WINDOW=0,LDU_BLK=0" > gpurun_out/r05u/probe_l12.log 2>&1
grep -v "^\[" gpurun_out/r05u/probe_l0.log | cut -c1-330; grep -v "^\[" gpurun_out/r05u/probe_l12.log | cut -c1-330
