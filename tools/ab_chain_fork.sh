for r in 1 2; do
 for v in fork serial; do
  if [ $v = fork ]; then A=""; else A="libchain_serial.so"; fi
  echo -n "$v serial-path: "; ALTLIB=$A python tools/exp_named_serial.py 128 2>&1 | grep -o "[0-9.]* ms per pair.*"
  echo -n "$v value: "; ALTLIB=$A python tools/bench_altlib.py --no-e2e --no-cpu-baseline --ragged-steps 0 --hard-steps 0 --resident-steps 0 --plan-check-pairs 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])"
 done
done
