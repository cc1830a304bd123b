for aw in 2 4 8 16; do for fd in 2 4; do
echo "== atomic_w=$aw fill_div=$fd"; DFINE_WGRAD_ATOMIC_W=$aw DFINE_WGRAD_FILL_DIV=$fd python tools/wgrad_in_step.py 2>&1 | grep -v Warn | sed -n 2,8p
done; done
