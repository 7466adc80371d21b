// Rank-4 field (component, z, y, x): a 3-D 7-point operator per component, the component index entering the body.
// The leading dimension carries no access offsets: the HIP lowering peels it off and launches one rank-3 apply per
// component (run_apply_batched); component 0 lies outside apply.bounds and is copied through.
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [0, 0, 0, 0], ub = [3, 10, 12, 256]>
!t = !neptune_ir.temp<element = f64, bounds = #b, location = #l>
!f = !neptune_ir.field<element = f64, bounds = #b, location = #l>
module {
  neptune_ir.nonlinear_opdef @lapc : (!t) -> !t {
  ^bb0(%u: !t):
    %r = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [1, 1, 1, 1], ub = [3, 9, 11, 255]>} : (!t) -> !t {
      ^bb0(%c: index, %i: index, %j: index, %k: index, %a: !t):
        %v0 = neptune_ir.access %a[0, 0, 0, 0] : !t -> f64
        %v1 = neptune_ir.access %a[0, -1, 0, 0] : !t -> f64
        %v2 = neptune_ir.access %a[0, 1, 0, 0] : !t -> f64
        %v3 = neptune_ir.access %a[0, 0, -1, 0] : !t -> f64
        %v4 = neptune_ir.access %a[0, 0, 1, 0] : !t -> f64
        %v5 = neptune_ir.access %a[0, 0, 0, -1] : !t -> f64
        %v6 = neptune_ir.access %a[0, 0, 0, 1] : !t -> f64
        %s0 = arith.addf %v1, %v2 : f64
        %s1 = arith.addf %s0, %v3 : f64
        %s2 = arith.addf %s1, %v4 : f64
        %s3 = arith.addf %s2, %v5 : f64
        %s4 = arith.addf %s3, %v6 : f64
        %w = arith.index_cast %c : index to i64
        %wf = arith.sitofp %w : i64 to f64
        %six = arith.constant 6.0 : f64
        %m = arith.mulf %six, %v0 : f64
        %d = arith.subf %s4, %m : f64
        %o = arith.mulf %wf, %d : f64
        neptune_ir.yield %o : f64
    }
    neptune_ir.return %r : !t
  }
  func.func @entry(%out: memref<?x?x?x?xf64>, %in0: memref<?x?x?x?xf64>) -> memref<?x?x?x?xf64> {
    %fo = neptune_ir.wrap %out : memref<?x?x?x?xf64> -> !f
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?xf64> -> !f
    %t0 = neptune_ir.load %f0 : !f -> !t
    %y = neptune_ir.apply_nonlinear @lapc(%t0) : (!t) -> !t
    neptune_ir.store %y to %fo : !t to !f
    %res = neptune_ir.unwrap %fo : !f -> memref<?x?x?x?xf64>
    func.return %res : memref<?x?x?x?xf64>
  }
  func.func @norm2(%in0: memref<?x?x?x?xf64>) -> f64 {
    %f0 = neptune_ir.wrap %in0 : memref<?x?x?x?xf64> -> !f
    %t0 = neptune_ir.load %f0 : !f -> !t
    %y = neptune_ir.apply_nonlinear @lapc(%t0) : (!t) -> !t
    %s = neptune_ir.reduce %y {kind = "sum"} : !t -> f64
    func.return %s : f64
  }
}
