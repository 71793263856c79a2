export WUNET_H3=1
for o in 32 23 3 432; do echo "W $o"; WUNET_H3W_ORDER=$o timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c90-200; done
