// A function of the shape every current-style reference input has (test/smoke_tests/smoke_time_advance.mlir:53-84):
// @entry computes a stencil value (%ustar: the explicit reaction step, body = @kat_react of kat_smoke_1d.mlir, i.e.
// smoke_time_advance.mlir:59-70) and hands it to an IMPLICIT time_advance, which stays on the host solver path.
// The HIP lowering does not lower @entry; it outlines %ustar as the exported symbol entry__stencil_0(out, in), whose
// result on the driver input 1..16 is KAT-2 of kat_reference_smoke.json.  Authored for this repository from the
// golden copies of the apply bodies; the operators @ol_lap / @ol_A restate @kat_lap / the second apply of @ac_A.
#c = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [0], ub = [16]>
!t = !neptune_ir.temp<element = f64, bounds = #b, location = #c>
!f = !neptune_ir.field<element = f64, bounds = #b, location = #c>
module {
  neptune_ir.linear_opdef @ol_lap : (!t) -> !t {
  ^bb0(%x: !t):
    %y = neptune_ir.apply(%x) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t) -> !t {
    ^bb0(%i: index, %a: !t):
      %m = neptune_ir.access %a[-1] : !t -> f64
      %z = neptune_ir.access %a[0] : !t -> f64
      %p = neptune_ir.access %a[1] : !t -> f64
      %k2 = arith.constant 2.0 : f64
      %k100 = arith.constant 100.0 : f64
      %v0 = arith.mulf %k2, %z : f64
      %v1 = arith.subf %m, %v0 : f64
      %v2 = arith.addf %v1, %p : f64
      %v3 = arith.mulf %k100, %v2 : f64
      neptune_ir.yield %v3 : f64
    }
    neptune_ir.return %y : !t
  }
  neptune_ir.linear_opdef @ol_A : (!t) -> !t {
  ^bb0(%x: !t):
    %lapx = neptune_ir.apply_linear @ol_lap(%x) : (!t) -> !t
    %y = neptune_ir.apply(%x, %lapx) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t, !t) -> !t {
    ^bb0(%i: index, %xa: !t, %la: !t):
      %x0 = neptune_ir.access %xa[0] : !t -> f64
      %l0 = neptune_ir.access %la[0] : !t -> f64
      %alpha = arith.constant 1.0e-4 : f64
      %s = arith.mulf %alpha, %l0 : f64
      %o = arith.subf %x0, %s : f64
      neptune_ir.yield %o : f64
    }
    neptune_ir.return %y : !t
  }
  func.func @entry(%out: memref<?xf64>, %in: memref<?xf64>) -> memref<?xf64> {
    %fout = neptune_ir.wrap %out : memref<?xf64> -> !f
    %fin = neptune_ir.wrap %in : memref<?xf64> -> !f
    %u0 = neptune_ir.load %fin : !f -> !t
    %ustar = neptune_ir.apply(%u0) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!t) -> !t {
    ^bb0(%i: index, %a: !t):
      %u = neptune_ir.access %a[0] : !t -> f64
      %dt = arith.constant 1.0e-2 : f64
      %u2 = arith.mulf %u, %u : f64
      %u3 = arith.mulf %u2, %u : f64
      %r = arith.subf %u, %u3 : f64
      %d = arith.mulf %dt, %r : f64
      %o = arith.addf %u, %d : f64
      neptune_ir.yield %o : f64
    }
    %dt = arith.constant 1.0e-2 : f64
    %u1 = neptune_ir.time_advance %ustar, %dt {method = 2 : i32, system = @ol_A, solver = "gmres", tol = 1.0e-8, max_iters = 200} : !t, f64 -> !t
    neptune_ir.store %u1 to %fout : !t to !f
    %res = neptune_ir.unwrap %fout : !f -> memref<?xf64>
    func.return %res : memref<?xf64>
  }
}
