for i in 1 2 3 4 5 6; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $((29600+i)) tests/rccl_ws1_child.py --out /tmp/ws1_$i.json > /tmp/ws1_$i.out 2> /tmp/ws1_$i.err
  echo "run $i rc=$?"
done
for i in 1 2 3 4 5 6; do if ! [ -s /tmp/ws1_$i.json ]; then echo "==== failed run $i"; grep -v "^\s*$" /tmp/ws1_$i.err | grep -v "frame #" | head -40; break; fi; done
