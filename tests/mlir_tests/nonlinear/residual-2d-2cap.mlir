// 2-D nonlinear residual with two captures, the shape the reference's SNES path calls back into
// (lib/Runtime/PETSc/NeptunePETScRuntime.cpp:1303-1361: NL<2, Caps>::FormFunction -- dlsym of the lowered
// nonlinear_opdef, one rank-2 memref for the iterate x plus one per capture, result copied out and freed):
//   F(x; up, kappa) = x - up - dt * kappa * lap5(x) + dt * x^3      inside the domain
//   F(x; up, kappa) = x - up                                        on the rim (scf.if on the index arguments,
//                                                                   like smoke_time_advance_nonlinear.mlir:29-38)
// 48 x 256 f64, ops in the textual order below.  Authored for the MI355X backend.
#loc = #neptune_ir.location<"cell">
#b   = #neptune_ir.bounds<lb = [0, 0], ub = [48, 256]>
!temp = !neptune_ir.temp<element = f64, bounds = #b, location = #loc>
module {
  neptune_ir.nonlinear_opdef @residual : (!temp, !temp, !temp) -> !temp {
  ^bb0(%x: !temp, %up: !temp, %kappa: !temp):
    %f = neptune_ir.apply(%x, %up, %kappa) attributes {bounds = #neptune_ir.bounds<lb = [0, 0], ub = [48, 256]>}
      : (!temp, !temp, !temp) -> !temp {
      ^bb0(%i: index, %j: index, %xa: !temp, %ua: !temp, %ka: !temp):
        %c0 = arith.constant 0 : index
        %ci = arith.constant 47 : index
        %cj = arith.constant 255 : index
        %e0 = arith.cmpi eq, %i, %c0 : index
        %e1 = arith.cmpi eq, %i, %ci : index
        %e2 = arith.cmpi eq, %j, %c0 : index
        %e3 = arith.cmpi eq, %j, %cj : index
        %e01 = arith.ori %e0, %e1 : i1
        %e23 = arith.ori %e2, %e3 : i1
        %rim = arith.ori %e01, %e23 : i1
        %xc = neptune_ir.access %xa[0, 0] : !temp -> f64
        %uc = neptune_ir.access %ua[0, 0] : !temp -> f64
        %d = arith.subf %xc, %uc : f64
        %v = scf.if %rim -> (f64) {
          scf.yield %d : f64
        } else {
          %n = neptune_ir.access %xa[-1, 0] : !temp -> f64
          %s = neptune_ir.access %xa[1, 0] : !temp -> f64
          %w = neptune_ir.access %xa[0, -1] : !temp -> f64
          %e = neptune_ir.access %xa[0, 1] : !temp -> f64
          %k = neptune_ir.access %ka[0, 0] : !temp -> f64
          %four = arith.constant 4.0 : f64
          %dt = arith.constant 1.0e-2 : f64
          %t0 = arith.addf %n, %s : f64
          %t1 = arith.addf %t0, %w : f64
          %t2 = arith.addf %t1, %e : f64
          %t3 = arith.mulf %four, %xc : f64
          %lap = arith.subf %t2, %t3 : f64
          %kl = arith.mulf %k, %lap : f64
          %dkl = arith.mulf %dt, %kl : f64
          %x2 = arith.mulf %xc, %xc : f64
          %x3 = arith.mulf %x2, %xc : f64
          %dx3 = arith.mulf %dt, %x3 : f64
          %r0 = arith.subf %d, %dkl : f64
          %r1 = arith.addf %r0, %dx3 : f64
          scf.yield %r1 : f64
        }
        neptune_ir.yield %v : f64
      }
    neptune_ir.return %f : !temp
  }
}
