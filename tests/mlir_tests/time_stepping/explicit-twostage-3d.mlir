// Explicit time stepping whose rhs opdef is NOT a single apply (two chained applies: a 7-point
// Laplacian of a pointwise-scaled state), 3-D.  The HIP lowering cannot fuse rhs and axpy here and
// takes the two-kernel form: call @rhs, then the axpy apply (HighLevelConvertion.cpp:77-120 shape).
#l = #neptune_ir.location<"cell">
#b = #neptune_ir.bounds<lb = [0, 0, 0], ub = [10, 9, 128]>
!t = !neptune_ir.temp<element = f64, bounds = #b, location = #l>
!f = !neptune_ir.field<element = f64, bounds = #b, location = #l>
module {
  neptune_ir.nonlinear_opdef @rhs : (!t) -> !t {
  ^bb0(%u: !t):
    %s = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [0, 0, 0], ub = [10, 9, 128]>} : (!t) -> !t {
      ^bb0(%i: index, %j: index, %k: index, %a: !t):
        %c = neptune_ir.access %a[0, 0, 0] : !t -> f64
        %h = arith.constant 0.5 : f64
        %c2 = arith.mulf %c, %c : f64
        %v = arith.mulf %h, %c2 : f64
        neptune_ir.yield %v : f64
    }
    %r = neptune_ir.apply(%s) attributes {bounds = #neptune_ir.bounds<lb = [1, 1, 1], ub = [9, 8, 127]>} : (!t) -> !t {
      ^bb0(%i: index, %j: index, %k: index, %a: !t):
        %c = neptune_ir.access %a[0, 0, 0] : !t -> f64
        %xm = neptune_ir.access %a[-1, 0, 0] : !t -> f64
        %xp = neptune_ir.access %a[1, 0, 0] : !t -> f64
        %ym = neptune_ir.access %a[0, -1, 0] : !t -> f64
        %yp = neptune_ir.access %a[0, 1, 0] : !t -> f64
        %zm = neptune_ir.access %a[0, 0, -1] : !t -> f64
        %zp = neptune_ir.access %a[0, 0, 1] : !t -> f64
        %six = arith.constant 6.0 : f64
        %t0 = arith.addf %xm, %xp : f64
        %t1 = arith.addf %t0, %ym : f64
        %t2 = arith.addf %t1, %yp : f64
        %t3 = arith.addf %t2, %zm : f64
        %t4 = arith.addf %t3, %zp : f64
        %t5 = arith.mulf %six, %c : f64
        %t6 = arith.subf %t4, %t5 : f64
        neptune_ir.yield %t6 : f64
    }
    neptune_ir.return %r : !t
  }
  func.func @step(%out: memref<?x?x?xf64>, %in: memref<?x?x?xf64>) -> memref<?x?x?xf64> {
    %fo = neptune_ir.wrap %out : memref<?x?x?xf64> -> !f
    %fi = neptune_ir.wrap %in : memref<?x?x?xf64> -> !f
    %u0 = neptune_ir.load %fi : !f -> !t
    %dt = arith.constant 1.25e-1 : f64
    %u1 = neptune_ir.time_advance %u0, %dt {method = 0 : i32, rhs = @rhs} : !t, f64 -> !t
    neptune_ir.store %u1 to %fo : !t to !f
    %res = neptune_ir.unwrap %fo : !f -> memref<?x?x?xf64>
    func.return %res : memref<?x?x?xf64>
  }
}
