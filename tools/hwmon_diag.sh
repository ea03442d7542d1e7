# which sysfs files carry the package power / engine clock of the GPU this process runs on (bench.py PowerWatch)
cd $GRAFT_REPO_ROOT
timeout 120 python tools/bench_conv.py 16 288 288 128 128 --gn --res --gn-part --fast-act --iters 6000 --wino > /dev/null 2>&1 &
pid=$!
sleep 14
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print('torch device 0:', p.name, 'pci domain/bus/device', getattr(p, 'pci_domain_id', None), getattr(p, 'pci_bus_id', None), getattr(p, 'pci_device_id', None))
PY
for d in /sys/class/drm/card*/device; do
  echo "== $d -> $(readlink -f $d)"
  for h in $d/hwmon/hwmon*; do
    [ -d $h ] || continue
    echo "   $h name=$(cat $h/name 2>/dev/null)"
    for f in power1_average power1_input power1_cap power1_label freq1_input freq1_label freq2_input freq2_label; do
      [ -e $h/$f ] && echo "      $f = $(cat $h/$f 2>/dev/null)"
    done
  done
done
rocm-smi --showbus --showpower --showclocks 2>/dev/null | grep -iE "GPU\[|power|sclk|bus" | head -20
wait $pid
