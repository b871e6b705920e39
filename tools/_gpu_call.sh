for i in 1 2; do
TMDIFF_WGRAD_BIAS=0 python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bias via channel_sum', d['ms_per_step'], d['train_step']['frac_of_fp32_mfma_peak'])"
python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bias in wgrad      ', d['ms_per_step'], d['train_step']['frac_of_fp32_mfma_peak'])"
done
