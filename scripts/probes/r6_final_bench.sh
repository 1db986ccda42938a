mkdir -p gpurun_out/r6c
python3 bench.py > gpurun_out/r6c/bench_final.json 2> gpurun_out/r6c/bench_final.err
python3 bench.py --points 30000 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 > gpurun_out/r6c/bench_chair30k.json 2>/dev/null
python3 bench.py --scene nerfsyn/lego.yml --points 30000 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 > gpurun_out/r6c/bench_lego30k.json 2>/dev/null
python3 bench.py --scene nerfsyn/lego.yml --points 30000 --amp --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 > gpurun_out/r6c/bench_lego30k_amp.json 2>/dev/null
python3 scripts/bench_render.py 30000 200 gpurun_out/r6c/render_lego30k_800.json
python3 scripts/bench_render.py 30000 400 gpurun_out/r6c/render_lego30k_800_chunk400.json
sed -i 's#gpurun_out/r5b/tl_#gpurun_out/r6c/tl_#' scripts/probes/r5_timeline.sh
bash scripts/probes/r5_timeline.sh default
bash scripts/probes/r5_timeline.sh amp --amp
python3 -c "
import json
for n in ('bench_final','bench_chair30k','bench_lego30k','bench_lego30k_amp'):
    j=json.load(open('gpurun_out/r6c/%s.json'%n)); print(n, j['ms_per_step'], j['value'], (j.get('as_shipped_amp') or {}).get('ms_per_step'), (j.get('parity_fp32_rows') or {}).get('ms_per_step'), (j.get('throughput_mode_h1') or {}).get('ms_per_step'), (j.get('psnr_after_steps') or {}).get('eval_psnr_db'))
"
