// Known-answer inputs: the four 1-D apply bodies of the reference's smoke tests, restated as
// stand-alone opdefs (same ops, same order, same constants).  Sources, reference tree:
//   @kat_lap      test/smoke_tests/smoke_time_advance.mlir:13-29      (@ac_lap)
//   @kat_react    test/smoke_tests/smoke_time_advance.mlir:59-70      (%ustar apply inside @entry)
//   @kat_bs       test/smoke_tests/smoke_time_advance_bs.mlir:13-51   (@bs_A)
//   @kat_resid    test/smoke_tests/smoke_time_advance_nonlinear.mlir:21-75 (@ac_residual)
//   @kat_axpy     test/smoke_tests/smoke_time_advance.mlir:38-49      (2-input apply inside @ac_A)
#c = #neptune_ir.location<"cell">
#b16 = #neptune_ir.bounds<lb = [0], ub = [16]>
#b32 = #neptune_ir.bounds<lb = [0], ub = [32]>
!t16 = !neptune_ir.temp<element = f64, bounds = #b16, location = #c>
!t32 = !neptune_ir.temp<element = f64, bounds = #b32, location = #c>
module {
  neptune_ir.linear_opdef @kat_lap : (!t16) -> !t16 {
  ^bb0(%x: !t16):
    %y = neptune_ir.apply(%x) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t16) -> !t16 {
    ^bb0(%i: index, %a: !t16):
      %m = neptune_ir.access %a[-1] : !t16 -> f64
      %z = neptune_ir.access %a[0] : !t16 -> f64
      %p = neptune_ir.access %a[1] : !t16 -> f64
      %k2 = arith.constant 2.0 : f64
      %k100 = arith.constant 100.0 : f64
      %v0 = arith.mulf %k2, %z : f64
      %v1 = arith.subf %m, %v0 : f64
      %v2 = arith.addf %v1, %p : f64
      %v3 = arith.mulf %k100, %v2 : f64
      neptune_ir.yield %v3 : f64
    }
    neptune_ir.return %y : !t16
  }
  neptune_ir.nonlinear_opdef @kat_react : (!t16) -> !t16 {
  ^bb0(%x: !t16):
    %y = neptune_ir.apply(%x) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t16) -> !t16 {
    ^bb0(%i: index, %a: !t16):
      %u = neptune_ir.access %a[0] : !t16 -> f64
      %dt = arith.constant 1.0e-2 : f64
      %u2 = arith.mulf %u, %u : f64
      %u3 = arith.mulf %u2, %u : f64
      %r = arith.subf %u, %u3 : f64
      %d = arith.mulf %dt, %r : f64
      %o = arith.addf %u, %d : f64
      neptune_ir.yield %o : f64
    }
    neptune_ir.return %y : !t16
  }
  neptune_ir.linear_opdef @kat_bs : (!t32) -> !t32 {
  ^bb0(%x: !t32):
    %y = neptune_ir.apply(%x) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [31]>} : (!t32) -> !t32 {
    ^bb0(%i: index, %a: !t32):
      %m = neptune_ir.access %a[-1] : !t32 -> f64
      %z = neptune_ir.access %a[0] : !t32 -> f64
      %p = neptune_ir.access %a[1] : !t32 -> f64
      %k100 = arith.constant 100.0 : f64
      %k5 = arith.constant 5.0 : f64
      %ca = arith.constant 2.0e-2 : f64
      %cb = arith.constant 3.0e-2 : f64
      %cc = arith.constant -5.0e-2 : f64
      %dt = arith.constant 1.0e-2 : f64
      %k2 = arith.constant 2.0 : f64
      %v0 = arith.mulf %k2, %z : f64
      %v1 = arith.subf %m, %v0 : f64
      %v2 = arith.addf %v1, %p : f64
      %xx = arith.mulf %k100, %v2 : f64
      %v3 = arith.subf %p, %m : f64
      %x1 = arith.mulf %k5, %v3 : f64
      %v4 = arith.mulf %ca, %xx : f64
      %v5 = arith.mulf %cb, %x1 : f64
      %v6 = arith.addf %v4, %v5 : f64
      %v7 = arith.mulf %cc, %z : f64
      %l = arith.addf %v6, %v7 : f64
      %dl = arith.mulf %dt, %l : f64
      %o = arith.subf %z, %dl : f64
      neptune_ir.yield %o : f64
    }
    neptune_ir.return %y : !t32
  }
  neptune_ir.nonlinear_opdef @kat_resid : (!t16, !t16) -> !t16 {
  ^bb0(%un: !t16, %up: !t16):
    %f = neptune_ir.apply(%un, %up) attributes {bounds = #neptune_ir.bounds<lb = [0], ub = [16]>} : (!t16, !t16) -> !t16 {
    ^bb0(%i: index, %a: !t16, %b: !t16):
      %c0 = arith.constant 0 : index
      %c15 = arith.constant 15 : index
      %l = arith.cmpi eq, %i, %c0 : index
      %r = arith.cmpi eq, %i, %c15 : index
      %e = arith.ori %l, %r : i1
      %v = scf.if %e -> (f64) {
        %a0 = neptune_ir.access %a[0] : !t16 -> f64
        %b0 = neptune_ir.access %b[0] : !t16 -> f64
        %d = arith.subf %a0, %b0 : f64
        scf.yield %d : f64
      } else {
        %m = neptune_ir.access %a[-1] : !t16 -> f64
        %z = neptune_ir.access %a[0] : !t16 -> f64
        %p = neptune_ir.access %a[1] : !t16 -> f64
        %o = neptune_ir.access %b[0] : !t16 -> f64
        %k2 = arith.constant 2.0 : f64
        %k100 = arith.constant 100.0 : f64
        %dt = arith.constant 1.0e-2 : f64
        %eps = arith.constant 1.0e-2 : f64
        %w0 = arith.mulf %k2, %z : f64
        %w1 = arith.subf %m, %w0 : f64
        %w2 = arith.addf %w1, %p : f64
        %lap = arith.mulf %k100, %w2 : f64
        %z2 = arith.mulf %z, %z : f64
        %z3 = arith.mulf %z2, %z : f64
        %re = arith.subf %z, %z3 : f64
        %df = arith.mulf %eps, %lap : f64
        %rhs = arith.addf %df, %re : f64
        %dr = arith.mulf %dt, %rhs : f64
        %w3 = arith.subf %z, %o : f64
        %res = arith.subf %w3, %dr : f64
        scf.yield %res : f64
      }
      neptune_ir.yield %v : f64
    }
    neptune_ir.return %f : !t16
  }
  neptune_ir.linear_opdef @kat_axpy : (!t16, !t16) -> !t16 {
  ^bb0(%x: !t16, %g: !t16):
    %y = neptune_ir.apply(%x, %g) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t16, !t16) -> !t16 {
    ^bb0(%i: index, %a: !t16, %b: !t16):
      %a0 = neptune_ir.access %a[0] : !t16 -> f64
      %b0 = neptune_ir.access %b[0] : !t16 -> f64
      %al = arith.constant 1.0e-4 : f64
      %t = arith.mulf %al, %b0 : f64
      %o = arith.subf %a0, %t : f64
      neptune_ir.yield %o : f64
    }
    neptune_ir.return %y : !t16
  }
}
