"""The unit metadata of coefficient containers: ``UnitValidator`` (expui/UnitValidator.H, .cc) -- which unit TYPES
(length, mass, time, velocity, G and their aliases) and which unit NAMES per type a ``Coefs.setUnits`` call accepts, and
the canonical spelling each alias is stored under.  The tables below are the reference's dictionaries as data
(expui/UnitValidator.cc:41-205), the ``cm/s -> cmm/s`` entry included as it stands there."""
from __future__ import annotations

from typing import Dict, List, Tuple

_TYPES: Dict[str, str] = {}
for _canon, _aliases in (("length", ("length", "Length", "Len", "len", "l", "L")),
                         ("mass", ("mass", "Mass", "m", "M")),
                         ("time", ("time", "Time", "t", "T")),
                         ("velocity", ("velocity", "vel", "Vel", "Velocity", "v", "V")),
                         ("G", ("G", "Grav", "grav", "grav_constant", "Grav_constant", "gravitational_constant",
                                "Gravitational_constant"))):
    for _a in _aliases:
        _TYPES[_a] = _canon

_UNITS: Dict[str, Dict[str, str]] = {
    "length": {"none": "none", "None": "none",
               "m": "m", "cm": "cm", "km": "km", "um": "um", "nm": "nm", "Angstrom": "Angstrom", "AU": "AU", "ly": "ly",
               "pc": "pc", "kpc": "kpc", "Mpc": "Mpc",
               "meter": "m", "centimeter": "cm", "kilometer": "km", "nanometer": "nm", "micrometer": "um", "micron": "um",
               "angstrom": "Angstrom", "AA": "Angstrom", "astronomical_unit": "AU", "au": "AU", "light_year": "ly",
               "lyr": "ly", "parsec": "pc", "kiloparsec": "kpc", "megaparsec": "Mpc"},
    "mass": {"none": "none", "None": "none", "Msun": "Msun", "Mearth": "Mearth", "g": "g", "kg": "kg",
             "solar_mass": "Msun", "earth_mass": "Mearth", "gram": "g", "kilograms": "kg"},
    "time": {"none": "none", "None": "none", "s": "s", "min": "min", "hr": "hr", "day": "day", "yr": "yr", "Myr": "Myr",
             "Gyr": "Gyr", "second": "s", "minute": "min", "hour": "hr", "year": "yr"},
    "velocity": {"none": "none", "cm/s": "cmm/s", "m/s": "m/s", "km/s": "km/s", "km/hr": "km/hr", "km/min": "km/min",
                 "c": "c", "meter_per_second": "m/s", "centimeter_per_second": "cm/s", "cm_per_s": "cm/s",
                 "m_per_s": "m/s", "km_per_s": "km/s", "km_per_hr": "km/hr", "km_per_min": "km/min",
                 "speed_of_light": "c"},
    "G": {"": "none", "mixed": "mixed", "none": "none", "unitless": "none"},
}


class UnitValidator:
    """``UnitValidator::operator()``: (valid, canonical type, canonical unit); an unknown type, or a unit the type does
    not list, gives (False, "unknown", "unknown")."""

    def __call__(self, type_: str, unit: str) -> Tuple[bool, str, str]:
        canon = _TYPES.get(type_)
        if canon is not None and unit in _UNITS[canon]:
            return True, canon, _UNITS[canon][unit]
        return False, "unknown", "unknown"

    @staticmethod
    def getAllowedTypes() -> List[str]:
        return ["mass", "length", "time", "velocity", "G"]

    @staticmethod
    def getAllowedTypeAliases(type_: str) -> List[str]:
        return sorted(a for a, c in _TYPES.items() if c == type_)

    @staticmethod
    def getAllowedUnits(type_: str) -> List[str]:
        return list(_UNITS.get(type_, {}))
