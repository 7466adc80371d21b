// Explicit time stepping through the stencil path: neptune_ir.time_advance {method = 0 (explicit), rhs = @lap}.
// Authored for the MI355X backend (SURVEY.md 8f rank 3); the reference lowers this method to
// apply_linear + an axpy apply (lib/Passes/HighLevelConvertion.cpp:77-120).
// u1 = u0 + dt * lap(u0); cells outside the operator's bounds see rhs = copy-through of the state.
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [0, 0], ub = [12, 128]>
!t = !neptune_ir.temp<element = f64, bounds = #b, location = #l>
!f = !neptune_ir.field<element = f64, bounds = #b, location = #l>
module {
  neptune_ir.linear_opdef @lap : (!t) -> !t {
  ^bb0(%u: !t):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [1, 1], ub = [11, 127]>} : (!t) -> !t {
      ^bb0(%i: index, %j: index, %a: !t):
        %c = neptune_ir.access %a[0, 0] : !t -> f64
        %n = neptune_ir.access %a[-1, 0] : !t -> f64
        %s = neptune_ir.access %a[1, 0] : !t -> f64
        %w = neptune_ir.access %a[0, -1] : !t -> f64
        %e = neptune_ir.access %a[0, 1] : !t -> f64
        %four = arith.constant 4.0 : f64
        %t0 = arith.addf %n, %s : f64
        %t1 = arith.addf %t0, %w : f64
        %t2 = arith.addf %t1, %e : f64
        %t3 = arith.mulf %four, %c : f64
        %t4 = arith.subf %t2, %t3 : f64
        neptune_ir.yield %t4 : f64
    }
    neptune_ir.return %r : !t
  }
  func.func @step(%out: memref<?x?xf64>, %in: memref<?x?xf64>) -> memref<?x?xf64> {
    %fo = neptune_ir.wrap %out : memref<?x?xf64> -> !f
    %fi = neptune_ir.wrap %in : memref<?x?xf64> -> !f
    %u0 = neptune_ir.load %fi : !f -> !t
    %dt = arith.constant 1.0e-1 : f64
    %u1 = neptune_ir.time_advance %u0, %dt {method = 0 : i32, rhs = @lap} : !t, f64 -> !t
    neptune_ir.store %u1 to %fo : !t to !f
    %res = neptune_ir.unwrap %fo : !f -> memref<?x?xf64>
    func.return %res : memref<?x?xf64>
  }
}
