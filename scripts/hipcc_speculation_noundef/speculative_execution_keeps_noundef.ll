target triple = "amdgcn-amd-amdhsa"
declare i32 @llvm.umin.i32(i32, i32)

define i32 @f(i32 %U, i32 %p) {
entry:
  %cmp = icmp ult i32 %p, %U
  br i1 %cmp, label %then, label %join

then:
  %sub = sub nuw i32 %U, %p
  %m = call noundef i32 @llvm.umin.i32(i32 4, i32 %sub)
  br label %join

join:
  %r = phi i32 [ %m, %then ], [ 0, %entry ]
  ret i32 %r
}
