export TMPDIR=/tmp
O=gpurun_out/r04k; mkdir -p $O
tools/lr_driver_profile.sh $O/lr 8
for i in 1 2 3 4 5 6 7 8; do echo "--- run $i"; cat $O/lr/run$i.txt | grep -v "operations, [134] dep" ; done
