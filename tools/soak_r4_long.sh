cd /root/repo
for t in "fuzz_batch.py 300 141" "fuzz_parity.py 1000 142" "fuzz_large_n.py 143 150" "fuzz_api_order.py 144 3000" "fuzz_strips_policy.py 145 40" "fuzz_rows.py 600 146" "fuzz_batch_rows.py 147 120" "fuzz_pyramid.py 148 120"; do
  echo "=== $t"; ( time timeout 1500 python tools/$t 2>&1 | grep -E "disagree|DISAGREE|Error|error|Traceback" | tail -6 ) 2>&1 | grep -v "^$" | tail -8
done
