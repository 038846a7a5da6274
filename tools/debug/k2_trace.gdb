# Developer probe (GPU box): trace the LDS mailbox (lanes 0..2, both slots) at entry and return of every Fq3-753 mailbox product / square
# inside the failing merge_like<G2Cfg3SMB, false> launch of build/k2_mailbox_check_fence.
#   rocgdb -batch -x tools/debug/k2_trace.gdb build/k2_mailbox_check_fence > trace.log ;  python tools/debug/k2_trace_check.py trace.log
set pagination off
set confirm off
set breakpoint pending on
define dumpmb
  set $k = 0
  while $k < 14
    eval "x/12xw local#%d", $k * 1024
    set $k = $k + 1
  end
end
break _Z10merge_likeIN3pcd9G2Cfg3SMBINS0_5F753BENS0_5F753AELj11ELj11ELi3EEELb0EEvPjj
run
delete 1
break _ZN3pcd4Fp3SINS_2FpINS_5F753BELb0ELb1EEELj11EE11mb_mul_callEPU3AS3Dv4_j
commands
  silent
  printf "@@ ENTER mul ret=%#lx\n", ((unsigned long)$s31 << 32) | (unsigned int)$s30
  dumpmb
  continue
end
break *((unsigned long)&_ZN3pcd4Fp3SINS_2FpINS_5F753BELb0ELb1EEELj11EE11mb_mul_callEPU3AS3Dv4_j + 0x8168)
commands
  silent
  printf "@@ LEAVE mul ret=%#lx\n", ((unsigned long)$s31 << 32) | (unsigned int)$s30
  dumpmb
  continue
end
break _ZN3pcd4Fp3SINS_2FpINS_5F753BELb0ELb1EEELj11EE11mb_sqr_callEPU3AS3Dv4_j
commands
  silent
  printf "@@ ENTER sqr ret=%#lx\n", ((unsigned long)$s31 << 32) | (unsigned int)$s30
  dumpmb
  continue
end
break *((unsigned long)&_ZN3pcd4Fp3SINS_2FpINS_5F753BELb0ELb1EEELj11EE11mb_sqr_callEPU3AS3Dv4_j + 0x6790)
commands
  silent
  printf "@@ LEAVE sqr ret=%#lx\n", ((unsigned long)$s31 << 32) | (unsigned int)$s30
  dumpmb
  continue
end
printf "@@ KERNEL base=%#lx\n", (unsigned long)&_Z10merge_likeIN3pcd9G2Cfg3SMBINS0_5F753BENS0_5F753AELj11ELj11ELi3EEELb0EEvPjj
continue
quit
