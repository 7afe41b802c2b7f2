"""The condensers of the rocprofv3 counter passes (tools/pmc_mfma.py, tools/pmc_sq.py) on a hand-made counter CSV."""
import csv
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
SYNTH = "void synth_mfma_kernel<unsigned short, unsigned short, true, true, 4, 4>(unsigned short const*, float const*)"
GRADV = "void grad_v_f32_kernel<2, 16>(float const*, float const*, float*, int, int, int, int, int, int)"


def _write(tmp_path, rows):
    d = tmp_path / "pass" / "host"
    d.mkdir(parents=True)
    with open(d / "1_counter_collection.csv", "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=FIELDS)
        w.writeheader()
        for r in rows:
            w.writerow(dict(zip(FIELDS, r)))
    return str(tmp_path / "pass")


def _run(tool, d):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), d], capture_output=True, text=True, check=True,
                          env={**os.environ, "K": "50"}).stdout


def test_mfma_utilisation_table(tmp_path):
    rows = []
    for i in range(4):                                   # 100 us launches at 2.0 GHz: GUI_ACTIVE = 8 XCDs x 200 000 cycles
        rows += [(i, GRADV, "SQ_VALU_MFMA_BUSY_CYCLES", 57802752, 1000 * i, 1000 * i + 100000),
                 (i, GRADV, "GRBM_GUI_ACTIVE", 1600000, 1000 * i, 1000 * i + 100000)]
    rows += [(9, "void at::native::vectorized_elementwise_kernel<4>()", "SQ_VALU_MFMA_BUSY_CYCLES", 0, 0, 10)]
    out = _run("pmc_mfma.py", _write(tmp_path, rows))
    line = [l for l in out.splitlines() if l.startswith("| `grad_v_f32_kernel<2, 16>`")]
    assert len(line) == 1 and "vectorized_elementwise" not in out
    cells = [c.strip() for c in line[0].split("|")]
    assert cells[2] == "4" and cells[3] == "100.0"
    assert cells[5] == "1.806 M"                         # 57 802 752 / 32 MFMAs
    assert cells[6] == "2.00 GHz"
    assert cells[7] == "28.2 %"                          # 57.8 M / (1024 SIMDs x 200 000 cycles)


def test_lds_conflicts_and_stall_split(tmp_path):
    vals = {"SQ_LDS_BANK_CONFLICT": 84, "SQ_LDS_IDX_ACTIVE": 216, "SQ_WAVE_CYCLES": 1000, "SQ_WAIT_ANY": 420,
            "SQ_WAIT_INST_ANY": 480, "SQ_ACTIVE_INST_ANY": 100, "SQ_ACTIVE_INST_VALU": 60, "SQ_ACTIVE_INST_LDS": 30}
    rows = [(0, GRADV, k, v, 0, 10) for k, v in vals.items()]
    rows += [(1, SYNTH, "SQ_WAVE_CYCLES", 500, 0, 10), (1, SYNTH, "SQ_WAIT_INST_ANY", 345, 0, 10)]
    out = _run("pmc_sq.py", _write(tmp_path, rows))
    gv = [l for l in out.splitlines() if l.startswith("| `grad_v_f32_kernel<2, 16>`")][0]
    assert "| 38.9 % | 42 % | 48 % | 10 % | 6 % | 3 % |" in gv
    sy = [l for l in out.splitlines() if l.startswith("| `synth_mfma_kernel<bf16, bf16, 1, 1, 4, 4>`")][0]
    assert "no LDS" in sy and "| 69 % |" in sy
