"""MI355X-native dense-tableau simplex core behind the YALPS solve()/Model/Solution API."""
