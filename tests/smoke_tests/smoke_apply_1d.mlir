// 1-D smoke input for tests/smoke_tests/smoke_apply_hip.sh: the reference's @ac_lap operator
// (test/smoke_tests/smoke_time_advance.mlir:13-29) behind the @entry shape of its smoke_apply.mlir
// (wrap, wrap, load, apply_linear, store, unwrap, return), with interior bounds so no access leaves the field.
#loc = #neptune_ir.location<"cell">
#b   = #neptune_ir.bounds<lb = [0], ub = [16]>
!temp  = !neptune_ir.temp<element = f64, bounds = #b, location = #loc>
!field = !neptune_ir.field<element = f64, bounds = #b, location = #loc>
module {
  neptune_ir.linear_opdef @A : (!temp) -> !temp {
  ^bb0(%u: !temp):
    %lap = neptune_ir.apply(%u) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [15]>} : (!temp) -> !temp {
      ^bb0(%i: index, %u_in: !temp):
        %um1 = neptune_ir.access %u_in[-1] : !temp -> f64
        %u0  = neptune_ir.access %u_in[0]  : !temp -> f64
        %up1 = neptune_ir.access %u_in[1]  : !temp -> f64
        %two    = arith.constant 2.0 : f64
        %dxinv2 = arith.constant 100.0 : f64
        %t0 = arith.mulf %two, %u0 : f64
        %t1 = arith.subf %um1, %t0 : f64
        %t2 = arith.addf %t1, %up1 : f64
        %lap_i = arith.mulf %dxinv2, %t2 : f64
        neptune_ir.yield %lap_i : f64
    }
    neptune_ir.return %lap : !temp
  }
  func.func @entry(%arg0: memref<?xf64>, %arg1: memref<?xf64>) -> memref<?xf64> {
    %f0 = neptune_ir.wrap %arg0 : memref<?xf64> -> !field
    %f1 = neptune_ir.wrap %arg1 : memref<?xf64> -> !field
    %t0 = neptune_ir.load %f1 : !field -> !temp
    %y  = neptune_ir.apply_linear @A(%t0) : (!temp) -> !temp
    neptune_ir.store %y to %f0 : !temp to !field
    %res = neptune_ir.unwrap %f0 : !field -> memref<?xf64>
    return %res : memref<?xf64>
  }
}
