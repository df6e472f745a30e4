#!/usr/bin/env python3
"""profiles/pmc_latest.json from a tools/pmc_passes.sh summary: per-launch HBM traffic of the proposal kernel with the
gfx950 FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section), MFMA pipe utilisation, VALU per MFMA.
usage: pmc_to_latest.py <pmc_summary.json> <tokens_per_launch> <source-note>"""
import hashlib, json, os, sys

summ = json.load(open(sys.argv[1]))
tokens = int(sys.argv[2])
name = next(k for k in summ if 'coarse_kernel' in k)
c = summ[name]
fetch_kb, write_kb = c['FETCH_SIZE'], c['WRITE_SIZE']
out = {
    'coarse_kernel_hbm_bytes_per_launch': 2 * fetch_kb * 1024 + write_kb * 1024,
    'tokens_per_launch': tokens,
    'fetch_size_kb_raw': fetch_kb, 'write_size_kb_raw': write_kb,
    'correction': 'bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request, '
                  'MI355X_MICROARCH.md HBM section; Infinity-Cache hits are included)',
    'mfma_busy_cycles': c.get('SQ_VALU_MFMA_BUSY_CYCLES'), 'grbm_gui_active_sum8xcd': c.get('GRBM_GUI_ACTIVE'),
    'insts_mfma': c.get('SQ_INSTS_MFMA'), 'insts_valu': c.get('SQ_INSTS_VALU'),
    'lds_bank_conflict': c.get('SQ_LDS_BANK_CONFLICT'),
    'wave_cycles': c.get('SQ_WAVE_CYCLES'), 'wait_any': c.get('SQ_WAIT_ANY'), 'wait_inst_any': c.get('SQ_WAIT_INST_ANY'),
    'active_inst_any': c.get('SQ_ACTIVE_INST_ANY'),
    'kernel': name, 'workload': f'bench.py default: N={tokens} tokens per launch, K=16384, D=256',
    'source': sys.argv[3] if len(sys.argv) > 3 else sys.argv[1],
    # the library the counters were collected with: bench.py reports `roofline.traffic` only for this very build
    'lib_sha256': hashlib.sha256(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'vector_quantization_amd', 'libvqhip.so'), 'rb').read()).hexdigest(),
}
if out['mfma_busy_cycles'] and out['grbm_gui_active_sum8xcd']:
    # SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs; GRBM_GUI_ACTIVE sums the 8 XCDs
    out['mfma_pipe_util'] = out['mfma_busy_cycles'] / (out['grbm_gui_active_sum8xcd'] / 8 * 1024)
if out['insts_mfma'] and out['insts_valu']:
    out['valu_per_mfma'] = out['insts_valu'] / out['insts_mfma']
if c.get('TCC_HIT_sum') is not None and c.get('TCC_MISS_sum') is not None:
    out['tcc_hit_rate'] = c['TCC_HIT_sum'] / max(1.0, c['TCC_HIT_sum'] + c['TCC_MISS_sum'])
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles', 'pmc_latest.json'), 'w'), indent=1)
print(json.dumps(out, indent=1))
