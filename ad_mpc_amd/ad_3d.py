"""Vehicle parameter + state holder with the interface of the reference's ``AD3D``
(data_driven_mpc/ros_gp_mpc/src/ad_mpc/ad_3d.py:20-105).  Values come from ad_mpc_amd.config, which
cites the reference lines."""
import numpy as np

from . import config as _c


class AD3D:
    def __init__(self, noisy=False, noisy_input=False):
        # state, one-element arrays as in the reference (ad_3d.py:39-45)
        for name in ("p_x", "p_y", "psi", "v_x", "v_y", "psi_dot", "delta"):
            setattr(self, name, np.zeros((1,)))
        # parameters (ad_3d.py:47-71)
        self.mass = _c.VEH_MASS
        self.f_mass = _c.VEH_F_MASS
        self.r_mass = _c.VEH_R_MASS
        self.L = _c.VEH_L
        self.L_F = _c.VEH_L_F
        self.L_R = _c.VEH_L_R
        self.Iz = _c.VEH_IZ
        self.Cf = _c.VEH_CF
        self.Cr = _c.VEH_CR
        self.blend_max = _c.BLEND_MAX
        self.blend_min = _c.BLEND_MIN
        self.steering_min, self.steering_max = _c.STEERING_MIN, _c.STEERING_MAX
        self.steering_rate_min, self.steering_rate_max = _c.STEERING_RATE_MIN, _c.STEERING_RATE_MAX
        self.acc_min, self.acc_max = _c.ACC_MIN, _c.ACC_MAX
        self.noisy_input = False
        self.noisy = noisy
        self.u_noiseless = np.array([0.0, 0.0])
        self.u = np.array([0.0, 0.0])

    _ORDER = ("p_x", "p_y", "psi", "v_x", "v_y", "psi_dot", "delta")

    def set_state(self, *args, **kwargs):
        """set_state([p_x,p_y,psi,v_x,v_y,psi_dot,delta]) or keyword form (ad_3d.py:83-95)."""
        if len(args) != 0:
            assert len(args) == 1 and len(args[0]) == 7
            for name, v in zip(self._ORDER, args[0]):
                getattr(self, name)[0] = v
        else:
            for name in self._ORDER:
                setattr(self, name, kwargs[name])

    def get_state(self, stacked=False):
        """ad_3d.py:97-101: flat list of 7 floats if stacked else list of the 7 one-element arrays."""
        if stacked:
            return [getattr(self, n)[0] for n in self._ORDER]
        return [getattr(self, n) for n in self._ORDER]

    def get_control(self, noisy=False):
        return self.u if noisy else self.u_noiseless
