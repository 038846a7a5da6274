# second probe: follow limb 6 of Y3 = r (V - X3) - 2 S1 J through the tail of the failing kernel (see tools/debug/k2_trace.gdb)
set pagination off
set confirm off
set breakpoint pending on
break _Z10merge_likeIN3pcd9G2Cfg3SMBINS0_5F753BENS0_5F753AELj11ELj11ELi3EEELb0EEvPjj
run
delete 1
printf "@@ kernel entry pc=%#lx\n", (unsigned long)$pc
set $K = (unsigned long)$pc
define show3
  printf "%s: %#x %#x %#x\n", $arg0, $arg1[0], $arg1[1], $arg1[2]
end
break *($K + 0x470c)
commands
  silent
  printf "@@ A after t[6] = r12[6] - D[6]\n"
  show3 "v34=r12[6]" $v34
  show3 "v20=D[6]" $v20
  show3 "v124=t[6]" $v124
  show3 "v125=t[5]" $v125
  show3 "v123=t[7]" $v123
  continue
end
break *($K + 0x4e8c)
commands
  silent
  printf "@@ B before call 14\n"
  show3 "v124" $v124
  p/x $s40
  p/x $s41
  continue
end
break *($K + 0x4e90)
commands
  silent
  printf "@@ C after call 14\n"
  show3 "v124" $v124
  p/x $s40
  p/x $s41
  continue
end
break *($K + 0x55c0)
commands
  silent
  printf "@@ D at the chain step of limb 6\n"
  show3 "v80=masked 2p[6]" $v80
  show3 "v18=2p[6]" $v18
  show3 "v124=t[6]" $v124
  show3 "v71=carry in" $v71
  p/x $s40
  p/x $s41
  stepi
  show3 "v71=sum" $v71
  x/12i $pc
  continue
end
continue
quit
