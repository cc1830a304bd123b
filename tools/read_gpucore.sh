#!/bin/bash
# After a GPU memory fault: name the kernel(s) whose waves were resident, from the gpucore the runtime wrote into the cwd.
core=$(ls -t gpucore.* 2>/dev/null | head -1)
[ -z "$core" ] && { echo "no gpucore"; exit 0; }
echo "core: $core ($(stat -c %s $core) bytes)"
timeout -k 5 240 /opt/rocm/bin/rocgdb -batch -ex "set pagination off" -ex "info threads" /usr/bin/python3 "$core" 2>&1 | grep -v "^warning\|^\[New\|No such file" | cut -c1-400 | head -${1:-80}
