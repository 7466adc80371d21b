// 1-D input of tests/smoke_tests/smoke_apply_hip.sh.  Authored for this backend: a weighted second difference
// w = 0.25 * ((left + right) - (centre + centre)) over the interior of a 24-cell field, behind an @entry that
// stores the operator's result into its first argument and returns it (the calling convention the reference's
// smoke driver exercises, test/smoke_tests/smoke_apply.sh:39-50).  On the ramp the driver feeds in, every interior
// value is exactly 0 and the two end cells are copy-through.
#cells = #neptune_ir.location<"cell">
#line  = #neptune_ir.bounds<lb = [0], ub = [24]>
!row  = !neptune_ir.temp<element = f64, bounds = #line, location = #cells>
!grid = !neptune_ir.field<element = f64, bounds = #line, location = #cells>
module {
  neptune_ir.linear_opdef @second_difference : (!row) -> !row {
  ^bb0(%state: !row):
    %out = neptune_ir.apply(%state) attributes {bounds = #neptune_ir.bounds<lb = [1], ub = [23]>} : (!row) -> !row {
      ^bb0(%cell: index, %s: !row):
        %centre = neptune_ir.access %s[0] : !row -> f64
        %left   = neptune_ir.access %s[-1] : !row -> f64
        %right  = neptune_ir.access %s[1] : !row -> f64
        %quarter = arith.constant 0.25 : f64
        %sides  = arith.addf %left, %right : f64
        %twice  = arith.addf %centre, %centre : f64
        %diff   = arith.subf %sides, %twice : f64
        %w      = arith.mulf %quarter, %diff : f64
        neptune_ir.yield %w : f64
    }
    neptune_ir.return %out : !row
  }
  func.func @entry(%dst: memref<?xf64>, %src: memref<?xf64>) -> memref<?xf64> {
    %gdst = neptune_ir.wrap %dst : memref<?xf64> -> !grid
    %gsrc = neptune_ir.wrap %src : memref<?xf64> -> !grid
    %now  = neptune_ir.load %gsrc : !grid -> !row
    %next = neptune_ir.apply_linear @second_difference(%now) : (!row) -> !row
    neptune_ir.store %next to %gdst : !row to !grid
    %ret  = neptune_ir.unwrap %gdst : !grid -> memref<?xf64>
    func.return %ret : memref<?xf64>
  }
}
