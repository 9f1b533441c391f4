import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, sys
m = importlib.import_module("dealii-stfem_amd")
L = m.lib()
try:
    op = m.StokesMatrixFreeOperator((2,2,2))
    print("create without torch ok", op.n_velocity, op.n_pressure)
except Exception as e:
    print("create without torch failed:", e, L.stfem_stokes_last_hip_error())
ctx = m.MatrixFreeOperator(2, (2,2,2))
print("scalar ctx ok", ctx.n_dofs)
try:
    op = m.StokesMatrixFreeOperator((2,2,2))
    print("create after scalar ctx ok")
except Exception as e:
    print("create after scalar failed:", e, L.stfem_stokes_last_hip_error())
