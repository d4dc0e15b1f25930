"""Host-side parameter objects with the reference's names and attributes.

These are INPUT HOLDERS for the drop-in API: users construct and mutate them exactly as with the
reference (`Rocket()`, `LiquidMotor()`, `rocket.dry_mass *= 1.02` ...), and `flatten.py` turns
them into the POD `erpl_config` / per-sample SoA rows the HIP library consumes.  The per-step
physics (mass properties, aerodynamic coefficients, thrust, atmosphere) lives ONLY in the HIP
kernels (csrc/); nothing here evaluates it on the CPU.

What does run on the host, bit-for-bit as in the reference, is the input preparation that
SURVEY.md §8a-13 keeps host-side: motor Monte Carlo perturbation (motor.py:95-125, :171-186) and
wind-profile synthesis (environment.py:118-265) on NumPy's legacy MT19937 `RandomState`.
"""
import copy

import numpy as np

LBF = 4.44822  # N per lbf, as used by the reference motor defaults (motor.py:22-23, :132-133)


class Rocket:
    """Attributes of rocket.py:14-66 (defaults identical).  `cp_location` follows the Barrowman
    computation of rocket.py:68-103 and, like the reference, is evaluated once at construction."""

    _DEFAULTS = dict(
        length=7.62, diameter=0.219, nose_length=0.2, fin_span=0.2, fin_root_chord=0.20,
        fin_tip_chord=0.1, fin_count=4, fin_sweep_angle=0.0, fin_cant_angle=0.0,
        dry_mass=113.4, propellant_mass=63.5, center_of_mass_dry=5.8,
        Ixx_dry=45, Iyy_dry=971.9, Izz_dry=971.693,
        parachute_area=15.0, parachute_cd=2.0, parachute_deployment_altitude=500,
        power_off_drag_factor=1.2,
    )

    def __init__(self, name="Sounding Rocket"):
        self.name = name
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        self.reference_area = np.pi * (self.diameter / 2) ** 2
        self.reference_diameter = self.diameter
        self.Cd_data = {
            "mach": [0.0, 0.5, 0.8, 1.0, 1.2, 1.5, 2.0, 3.0],
            "cd0": [0.4, 0.42, 0.48, 0.65, 0.52, 0.45, 0.40, 0.38],
            "cda": [1.2, 1.25, 1.3, 1.4, 1.35, 1.25, 1.2, 1.15],
        }
        self.CP_shift_data = {
            "mach": [0.0, 0.8, 1.0, 1.2, 2.0, 3.0],
            "cp_shift": [0.0, -0.05, -0.1, -0.05, 0.0, 0.0],
        }
        self.cp_location = self._calculate_center_of_pressure()

    def _calculate_center_of_pressure(self):
        """Barrowman CP from the nose tip (rocket.py:68-103): nose CN=2 at 0.666 L_nose, fins with
        interference factor (1 + d/2s), CP of the fin set at the quarter MAC."""
        cn_nose, x_nose = 2.0, 0.666 * self.nose_length
        cr, ct, s = self.fin_root_chord, self.fin_tip_chord, self.fin_span
        area = 0.5 * (cr + ct) * s
        lam = ct / cr if cr != 0 else 0.0
        cn_fins = 2 * self.fin_count * (1 + self.diameter / (2 * s)) * (area / self.reference_area)
        mac = (2 / 3) * cr * (1 + lam + lam ** 2) / (1 + lam)
        y_bar = s * (1 + 2 * lam) / (3 * (1 + lam))
        x_fins = (self.length - cr) + y_bar * np.tan(self.fin_sweep_angle) + 0.25 * mac
        cn_total = cn_nose + 0.0 + cn_fins
        if cn_total > 0:
            return (cn_nose * x_nose + 0.0 * 0.0 + cn_fins * x_fins) / cn_total
        return self.length / 2


class LiquidMotor:
    """motor.py:128-186: pressure-fed liquid engine, F = F_vac - A_e * P_amb until burn-out."""

    def __init__(self, name="Liquid Motor", thrust_vacuum=2590 * LBF, thrust_sea_level=2290 * LBF,
                 mass_flow_rate=4.26, propellant_mass=63.5):
        self.name = name
        self.thrust_vacuum = thrust_vacuum
        self.thrust_sea_level = thrust_sea_level
        self.mass_flow_rate = mass_flow_rate
        self.propellant_mass = propellant_mass
        self.nozzle_exit_area = (self.thrust_vacuum - self.thrust_sea_level) / 101325.0
        self.burn_time = self.propellant_mass / self.mass_flow_rate
        self.total_impulse = self.thrust_vacuum * self.burn_time
        self.thrust_uncertainty = 0.05
        self.mass_flow_uncertainty = 0.03

    def perturb_for_monte_carlo(self, random_state=None):
        """Two normal draws, thrust then mass flow (motor.py:171-186)."""
        rs = random_state if random_state is not None else np.random.RandomState()
        k_thrust = rs.normal(1.0, self.thrust_uncertainty)
        k_flow = rs.normal(1.0, self.mass_flow_uncertainty)
        return LiquidMotor(self.name + "_perturbed",
                           thrust_vacuum=self.thrust_vacuum * k_thrust,
                           thrust_sea_level=self.thrust_sea_level * k_thrust,
                           mass_flow_rate=self.mass_flow_rate * k_flow,
                           propellant_mass=self.propellant_mass)


class SolidMotor:
    """motor.py:8-125: tabulated sea-level thrust curve + nozzle pressure correction."""

    CURVE_TIME = (0.0, 0.2, 0.5, 1.0, 2.0, 5.0, 8.0, 12.0, 14.0, 15.0)
    CURVE_SHAPE = (0.0, 2.2, 2.0, 1.8, 1.5, 1.2, 1.0, 0.8, 0.3, 0.0)

    def __init__(self, name="Solid Motor"):
        self.name = name
        self.total_impulse = 156297
        self.burn_time = 15.0
        self.propellant_mass = 63.5
        self.average_thrust = self.total_impulse / self.burn_time
        self.thrust_sea_level = 2290 * LBF
        self.thrust_vacuum = 2590 * LBF
        self.nozzle_exit_area = (self.thrust_vacuum - self.thrust_sea_level) / 101325.0
        self.thrust_curve_time = np.array(self.CURVE_TIME)
        self.thrust_curve_normalized = np.array(self.CURVE_SHAPE)
        self.thrust_curve_thrust = self.thrust_curve_normalized * self.average_thrust
        self.mass_flow_rate = 4.26
        self.exhaust_velocity = self.average_thrust / self.mass_flow_rate
        self.thrust_uncertainty = 0.05
        self.burn_time_uncertainty = 0.02
        self.total_impulse_uncertainty = 0.03

    def perturb_for_monte_carlo(self, random_state=None):
        """Three normal draws: thrust, burn time, impulse (motor.py:95-125).  The curve's time
        axis is not rescaled; mass flow and exit area scale with the thrust multiplier."""
        rs = random_state if random_state is not None else np.random.RandomState()
        m = SolidMotor(self.name + "_perturbed")
        k = rs.normal(1.0, self.thrust_uncertainty)
        m.thrust_curve_thrust = self.thrust_curve_thrust * k
        m.average_thrust = self.average_thrust * k
        m.thrust_sea_level = self.thrust_sea_level * k
        m.thrust_vacuum = self.thrust_vacuum * k
        m.burn_time = self.burn_time * rs.normal(1.0, self.burn_time_uncertainty)
        m.total_impulse = self.total_impulse * rs.normal(1.0, self.total_impulse_uncertainty)
        m.mass_flow_rate = 4.26 * k
        m.exhaust_velocity = m.average_thrust / m.mass_flow_rate
        m.nozzle_exit_area = self.nozzle_exit_area * k
        m._thrust_multiplier = k  # the kernel applies it to the shared curve (motor.py:105)
        return m


class StandardAtmosphere:
    """Constants of environment.py:13-24.  The piecewise T/P/rho model (environment.py:26-103,
    including its discontinuities at 25 km and 32 km) is evaluated inside the HIP kernels."""

    def __init__(self):
        self.sea_level_pressure = 101325.0
        self.sea_level_temperature = 288.15
        self.sea_level_density = 1.225  # never read by the path
        self.temperature_lapse_rate = 0.0065
        self.gas_constant = 287.053
        self.gravity = 9.80665
        self.gamma = 1.4
        self.troposphere_height = 11000.0
        self.stratosphere_height = 20000.0
        self.stratosphere_temp = 216.65



def knot_constants(wind_model, altitudes):
    """Per-knot turbulence sigma, AR(1) correlation and innovation sigma of a WindModel-shaped object
    (this package's or the reference's: only `turbulence_intensity` and `correlation_length` are read)
    (environment.py:161, :176-181, :189 == :242, :249-255)."""
    alt = [np.float64(a) for a in altitudes]
    sigma = [wind_model.turbulence_intensity * np.exp(-a / 2000.0) for a in alt]
    rho, innov = [None], [None]
    for i in range(1, len(alt)):
        dz = max(alt[i] - alt[i - 1], 1e-6)
        r = np.clip(np.exp(-dz / wind_model.correlation_length), 0.1, 0.95)
        rho.append(r)
        innov.append(sigma[i] * np.sqrt(max(1 - r ** 2, 0.01)))
    return sigma, rho, innov


class WindModel:
    """Wind-profile preparation of environment.py:110-265 (host-side input prep)."""

    def __init__(self):
        self.power_law_exponent = 0.14
        self.turbulence_intensity = 2.0
        self.correlation_length = 100.0

    # -- deterministic pieces ---------------------------------------------------------------
    def power_law_profile(self, altitude, reference_wind_speed, reference_altitude=10.0):
        """environment.py:118-123 (both branches are the same expression)."""
        return reference_wind_speed * (altitude / reference_altitude) ** self.power_law_exponent

    def load_wind_profile_from_csv(self, file_path):
        """environment.py:202-216: columns altitude,u,v[,w] -> (K,), (K,3)."""
        data = np.genfromtxt(file_path, delimiter=",", names=True)
        alt = data["altitude"]
        w = data["w"] if "w" in data.dtype.names else np.zeros_like(alt)
        return alt, np.vstack([data["u"], data["v"], w]).T

    def _knot_constants(self, altitudes):
        return knot_constants(self, altitudes)

    # -- stochastic profiles (legacy RandomState stream, draw order u, v, w per knot) -------
    def perturb_wind_profile(self, altitudes, base_profile, random_state=None):
        """AR(1) turbulence added to a baseline profile (environment.py:218-265)."""
        rs = random_state if random_state is not None else np.random.RandomState()
        base = np.asarray(base_profile)
        sigma, rho, innov = self._knot_constants(altitudes)
        out = np.zeros_like(base)
        s0 = sigma[0]
        out[0, 0] = base[0, 0] + rs.normal(0, s0)
        out[0, 1] = base[0, 1] + rs.normal(0, s0)
        out[0, 2] = base[0, 2] + rs.normal(0, s0 * 0.3)
        for i in range(1, len(sigma)):
            prev = out[i - 1] - base[i - 1]
            tu = rho[i] * prev[0] + rs.normal(0, innov[i])
            tv = rho[i] * prev[1] + rs.normal(0, innov[i])
            tw = rho[i] * prev[2] + rs.normal(0, innov[i] * 0.3)
            out[i, 0] = base[i, 0] + tu
            out[i, 1] = base[i, 1] + tv
            out[i, 2] = base[i, 2] + tw
        return out

    def generate_stochastic_profile(self, altitudes, base_wind_speed, base_wind_direction=None,
                                    random_state=None):
        """Power-law mean wind + AR(1) turbulence (environment.py:125-200)."""
        rs = random_state if random_state is not None else np.random.RandomState()
        if base_wind_direction is None:
            base_wind_direction = rs.uniform(0.0, 2 * np.pi)
        cd, sd = np.cos(base_wind_direction), np.sin(base_wind_direction)
        sigma, rho, innov = self._knot_constants(altitudes)
        mean = [self.power_law_profile(a, base_wind_speed) for a in altitudes]
        out = np.zeros((len(sigma), 3))
        out[0, 0] = mean[0] * cd + rs.normal(0, sigma[0])
        out[0, 1] = mean[0] * sd + rs.normal(0, sigma[0])
        out[0, 2] = rs.normal(0, sigma[0] * 0.3)
        for i in range(1, len(sigma)):
            pu = out[i - 1, 0] - mean[i - 1] * cd
            pv = out[i - 1, 1] - mean[i - 1] * sd
            pw = out[i - 1, 2]
            tu = rho[i] * pu + rs.normal(0, innov[i])
            tv = rho[i] * pv + rs.normal(0, innov[i])
            tw = rho[i] * pw + rs.normal(0, innov[i] * 0.3)
            out[i, 0] = mean[i] * cd + tu
            out[i, 1] = mean[i] * sd + tv
            out[i, 2] = tw
        return out


def clone(obj):
    """deepcopy used for per-sample perturbation (monte_carlo.py:311-312, :329-330)."""
    return copy.deepcopy(obj)
